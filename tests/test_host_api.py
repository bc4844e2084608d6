"""Host-side mirror of the reference API (no GPU needed for these checks) and the C-ABI library's
export table."""
import ctypes as C
import os
import re

import pytest

from compressjs_amd import Bzip2, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_invalid_block_size_multiplier_message():
    for lv in (0, 10, -1, 2.5):
        with pytest.raises(ValueError, match="Invalid block size multiplier"):   # lib/Bzip2.js:888-890
            Bzip2.compressFile(b"abc", None, lv)


def test_library_exports_every_declared_symbol():
    """The HIP library must load on a GPU-less machine and export all of include/*.h."""
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libcompressjs_amd.so not built (run __graft_entry__.build())")
    hdr = open(os.path.join(ROOT, "include", "compressjs_amd.h")).read()
    declared = set(re.findall(r"\b(cjs_[a-z0-9_]+)\s*\(", hdr))
    L = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert declared >= set(_lib.SYMBOLS)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(_lib.LIB_PATH):
        with pytest.raises(_lib.CompressjsAmdError):
            Bzip2.compressFile(b"abc")
        return
    with pytest.raises(_lib.CompressjsAmdError, match="no CPU path"):
        Bzip2.compressFile(b"abc")


def test_product_never_imports_the_oracle():
    pk = os.path.join(ROOT, "compressjs_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cc", ".js")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in src and "oracle/" not in src.replace("the oracle", ""), f


def test_node_dropin_loads_and_fails_loudly_without_gpu():
    import shutil
    import subprocess
    import torch
    if torch.cuda.is_available() or shutil.which("node") is None:
        pytest.skip("needs node and no GPU")
    if not os.path.exists(os.path.join(ROOT, "build", "compressjs_amd.node")):
        pytest.skip("addon not built")
    js = ("var c=require('./js');"
          "try{c.Bzip2.compressFile(Buffer.from('abc'));console.log('NOTHROW')}catch(e){console.log('E1:'+e.message)}"
          "try{c.Bzip2.compressFile(Buffer.from('abc'),null,0)}catch(e){console.log('E2:'+e.message)}")
    out = subprocess.check_output(["node", "-e", js], cwd=ROOT).decode()
    assert "no CPU path" in out and "E2:Invalid block size multiplier" in out and "NOTHROW" not in out


def test_bench_documents_tile_the_job_stream():
    """bench.py --gpus N: the job's stream is one document per rank; a rank's window (its document + the margin in front of it)
    must be exactly those bytes of the whole stream, and document 0 the single-GPU stream."""
    import workloads
    n, world = 200_000, 3
    for name in ("enwik", "text", "lcg") + (("e8sa",) if workloads.have_fixtures() else ()):
        whole = workloads.world_stream(name, n, world)
        assert whole.size == n * world and (whole[:n] == workloads.stream(name, n)).all()
        for r in range(world):
            for margin in (0, 999, n, n + 5, 2 * n + 7):
                lo = max(0, r * n - margin)
                assert (workloads.window(name, n, r, margin) == whole[lo:(r + 1) * n]).all(), (name, r, margin)


def test_range_coder_division_by_reciprocal_is_exact():
    """bwtc_host.hip replaces RangeCoder.encodeFreq's `range / tot` (lib/RangeCoder.js:81) by a multiplication with
    ceil(2^48 / tot): exact for tot <= 0xFFFF, range < 2^32.  Every tot against the ranges where a wrong floor would show:
    multiples of tot and their neighbours around every power of two, the ends of the range, and random ranges."""
    import numpy as np
    import stagelib
    L = C.CDLL(stagelib.build_emu())
    L.cjs_dbg_rc_div.restype = C.c_uint32
    L.cjs_dbg_rc_div.argtypes = [C.c_uint32, C.c_uint32]
    rng = np.random.RandomState(3)
    tots = list(range(1, 300)) + [int(x) for x in rng.randint(300, 65536, size=1500)] + [65535, 65534, 0xFF00, 0xFEFF, 32768, 32767, 32769]
    for tot in tots:
        cand = {0, 1, tot - 1, tot, tot + 1, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 0x7FFFFFFF, 0x00800000, 0x007FFFFF}
        for k in range(8, 33):
            q = (1 << k) // tot
            for m in (q - 1, q, q + 1):
                for d in (-1, 0, 1):
                    cand.add(m * tot + d)
        top = (0xFFFFFFFF // tot) * tot
        cand.update((top - 1, top, top + 1))
        cand.update(int(x) for x in rng.randint(0, 1 << 32, size=8, dtype=np.uint64))
        for r in cand:
            if 0 <= r <= 0xFFFFFFFF:
                assert L.cjs_dbg_rc_div(r, tot) == r // tot, (r, tot)
    assert L.cjs_dbg_rc_div(0xFFFFFFFF, 70000) == 0xFFFFFFFF // 70000      # beyond the table: plain division
