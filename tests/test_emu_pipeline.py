"""CPU logic tests of the HIP sources: the product's unmodified .hip files are compiled by g++
against the fake HIP runtime in tests/emu (fibers, 64-lane wave collectives) and every stage is
compared with the oracle.  These are NOT parity claims for the GPU build -- tests/test_gpu_*.py
are -- they keep the kernel logic and the host orchestration honest in a container without GPU."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import cases
import oracle
import stagelib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from compressjs_amd import _lib, synth


@pytest.fixture(scope="module")
def emu():
    L = stagelib.load("emu")
    return L


@pytest.fixture(scope="module")
def emu_ctx():
    L = _lib.load(stagelib.build_emu())
    h = L.cjs_create(0, 2)
    assert h
    yield L, h
    L.cjs_destroy(h)


def _compress(Lh, data, level):
    L, h = Lh
    d = np.ascontiguousarray(data, dtype=np.uint8)
    cap = int(L.cjs_bz2_compress_bound(d.size))
    out = np.zeros(cap, np.uint8)
    n = L.cjs_bz2_compress(h, d.ctypes.data, d.size, level, out.ctypes.data, cap)
    assert n >= 0, n
    return out[:n].tobytes()


SMALL = ["empty", "a1", "a3", "a4", "a5", "a8", "a255", "a256", "a259", "a260", "a1000", "ab500",
         "abc_tie", "banana", "bytes40", "mary9", "text1k", "sample0"]


@pytest.mark.parametrize("cid", SMALL)
def test_stream_small_cases_vs_reference_digest(emu_ctx, golden, cid):
    d = cases.case_input(cid)
    if d is None:
        pytest.skip("fixture not staged")
    for k in [k for k in golden if k.startswith(cid + ":bz2:")]:
        lv = int(k.split(":")[2])
        o = _compress(emu_ctx, d, lv)
        assert len(o) == golden[k]["out_len"], k
        assert hashlib.sha256(o).hexdigest() == golden[k]["out_sha256"], k


@pytest.mark.parametrize("cid,level", [("lcg99977_a10", 1), ("lcg99976_a300", 1), ("runs300k", 1),
                                       ("zeros300k", 9), ("text100k", 1)])
def test_stream_block_boundary_cases(emu_ctx, golden, cid, level):
    d = cases.case_input(cid)
    v = golden["%s:bz2:%d" % (cid, level)]
    o = _compress(emu_ctx, d, level)
    assert len(o) == v["out_len"] and hashlib.sha256(o).hexdigest() == v["out_sha256"]


def test_stage_by_stage_vs_oracle(emu):
    d = np.concatenate([synth.text_like(30000, 9), synth.runs_mixed(20000, 4), synth.lcg_ascii(15000, 3)])
    level, cap = 1, 99981
    orc = list(oracle.block_stages(d, level))
    dev = stagelib.block_stages(emu, [o["T"] for o in orc], cap, 5, crcs=[o["crc"] for o in orc], level=level)
    for dv, o in zip(dev, orc):
        assert stagelib.compare_with_oracle(dv, o, 5) == []
    assert dev[0]["stream"] == oracle.bz2_compress(d, level)


def test_selector_mtf_quirk_tiny_alphabet(emu_ctx):
    """alphabetSize < nGroups: the reference's selector MTF runs on a too-short Uint8Array
    (lib/Bzip2.js:850-862); the kernels mirror it.  Oracle is pinned on the same inputs."""
    rng = np.random.RandomState(5)
    words = [b"ab", b"aab", b"abb", b"ba"]
    d = np.frombuffer(b"".join(words[i] for i in rng.randint(0, 4, size=6000)), dtype=np.uint8)
    assert _compress(emu_ctx, d, 9) == oracle.bz2_compress(d, 9)


def test_multi_batch_equals_single(emu_ctx):
    d = synth.text_like(250000, 12)           # 3 blocks at level 1, batches of 2
    assert _compress(emu_ctx, d, 1) == oracle.bz2_compress(d, 1)


def test_long_runs_count_bytes_and_split_crc(emu_ctx):
    """Inputs of long runs: a block then consumes megabytes of input - k0_crc splits it over several workgroups (one part per 2 MB:
    the parts' raw remainders XOR into the constant term k0_pad left) and k0_materialize finds the count byte behind a run's fourth
    byte eight input bytes at a time (runs of every length 4..270 at every alignment, cut by block ends and by the end of the input)."""
    rng = np.random.RandomState(11)
    d = np.concatenate([np.zeros(5_000_000, np.uint8), synth.text_like(3000, 1), np.full(2_600_000, 65, np.uint8), np.zeros(7, np.uint8)])
    assert _compress(emu_ctx, d, 1) == oracle.bz2_compress(d, 1)
    parts = []
    for k in range(700):
        parts.append(np.full(int(rng.randint(4, 271)), int(rng.randint(0, 3)), np.uint8))
        parts.append(rng.randint(3, 256, size=int(rng.randint(0, 9))).astype(np.uint8))
    d = np.concatenate(parts)
    for cut in (d.size, d.size - 1, d.size - 5, 99981 + 3, 99981 + 4, 99981 + 5):
        assert _compress(emu_ctx, d[:cut], 1) == oracle.bz2_compress(d[:cut], 1), cut
    for off in range(1, 9):                                  # the same runs at other alignments of the input
        assert _compress(emu_ctx, d[off:60000 + off], 1) == oracle.bz2_compress(d[off:60000 + off], 1), off


def test_bwt_kats(emu):
    L = _lib.load(stagelib.EMU_SO)
    for inp, out, idx in [(b"bcababa", b"cbbaaab", 5), (b"abab", b"bbaa", 1), (b"aaaa", b"aaaa", 3),
                          (b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES",
                           b"TEXYDST.E.IXIXIXXSSMPPS.B..E.S.EUSFXDIIOIIIT", 29)]:
        t = np.frombuffer(inp, dtype=np.uint8).copy()
        u = np.zeros(t.size, np.uint8)
        p = C.c_uint32(0)
        assert L.cjs_bwt_cyclic(t.ctypes.data, u.ctypes.data, t.size, C.byref(p)) == 0
        assert u.tobytes() == out and p.value == idx


def test_lcg_generator_on_the_device(emu_ctx):
    """cjs_lcg_ascii_device (jump-ahead) = synth.lcg_ascii = SURVEY.md 8(c)'s LCG(n, seed), for any slice."""
    L, h = emu_ctx
    want = synth.lcg_ascii(70000, 7)
    for first, n in ((0, 70000), (1, 999), (4097, 12345), (69999, 1), (31, 0)):
        out = np.full(n + 1, 0xEE, np.uint8)
        assert L.cjs_lcg_ascii_device(h, out.ctypes.data, n, 7, first) == 0
        assert np.array_equal(out[:n], want[first:first + n]) and out[n] == 0xEE, (first, n)


def test_invalid_level_code(emu_ctx):
    L, h = emu_ctx
    d = np.zeros(4, np.uint8)
    out = np.zeros(256, np.uint8)
    assert L.cjs_bz2_compress(h, d.ctypes.data, 4, 0, out.ctypes.data, 256) == -20
    assert L.cjs_bz2_compress(h, d.ctypes.data, 4, 10, out.ctypes.data, 256) == -20


def test_doubling_rounds_of_suffix_sort():
    """Passages repeated far apart outlast the text stages: the list-driven doubling rounds (k1_dbl.hip) must run for several
    rounds (counted with CJS_K1_TRACE, which reads the lists' counters back) and give the oracle's transform."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib
from compressjs_amd import synth
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic.restype = C.c_int32
L.cjs_bwt_cyclic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
passage = synth.text_like(700, 3)
d = np.concatenate([synth.text_like(20000, 1), passage, synth.text_like(15000, 2), passage,
                    synth.runs_mixed(3000, 1), passage, synth.text_like(9000, 4)])
u = np.zeros(d.size, np.uint8); p = C.c_uint32(0)
assert L.cjs_bwt_cyclic(d.ctypes.data, u.ctypes.data, d.size, C.byref(p)) == 0
uo, po = oracle.bwt_cyclic(d)
assert p.value == po and (u == uo).all()
assert L.cjs_dbg_k1_sparse_rounds() >= 3, L.cjs_dbg_k1_sparse_rounds()
print("ok", L.cjs_dbg_k1_rounds(), L.cjs_dbg_k1_sparse_rounds())
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    env = dict(os.environ, CJS_K1_TRACE="1")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600, stderr=subprocess.DEVNULL)
    assert out.decode().startswith("ok")


def test_linear_bwt_and_suffix_array(golden):
    """BWT.bwtransform / BWT.suffixsort (implicit smallest sentinel) on the linear mode of K1,
    against the reference-made vectors and the oracle."""
    L = _lib.load(stagelib.EMU_SO)
    for cid in ["sample0", "banana", "a1", "a4", "ab500", "abc_tie", "mary9", "bytes40", "text1k"]:
        d = cases.case_input(cid)
        if d is None:
            continue
        d = np.ascontiguousarray(d)
        u = np.zeros(max(d.size, 1), np.uint8)
        p = C.c_uint32(0)
        assert L.cjs_bwt_linear(d.ctypes.data, u.ctypes.data, d.size, C.byref(p)) == 0
        sa = np.zeros(max(d.size, 1), np.int32)
        assert L.cjs_suffixsort(d.ctypes.data, sa.ctypes.data, d.size) == 0
        v = golden[cid + ":bwt"]
        assert p.value == v["pidx"] and hashlib.sha256(u[:d.size].tobytes()).hexdigest() == v["u_sha256"], cid
        assert hashlib.sha256(sa[:d.size].astype("<i4").tobytes()).hexdigest() == golden[cid + ":sa"]["sa_sha256"], cid
    for raw in (b"ab\0\0\0\0\0\0\0\0\0ab\0\0\0", b"\0" * 50, b"mississippi"):
        d = np.frombuffer(raw, dtype=np.uint8).copy()
        u = np.zeros(d.size, np.uint8)
        p = C.c_uint32(0)
        assert L.cjs_bwt_linear(d.ctypes.data, u.ctypes.data, d.size, C.byref(p)) == 0
        uo, po = oracle.bwt_linear(d)
        assert p.value == po and np.array_equal(u, uo)


def test_bwt_entry_points_above_one_megabyte():
    """BWT.bwtransform / suffixsort on a block of more than 2^20 bytes (the reference's test/bwtest.js and suftest.js
    go up to sample5.ref = 2 130 640): ranks packed into 22 bits, most buckets of the front end beyond LDS (task levels)."""
    L = _lib.load(stagelib.EMU_SO)
    d = np.ascontiguousarray(synth.text_like((1 << 20) + 70_001, 9))
    u = np.zeros(d.size, np.uint8)
    p = C.c_uint32(0)
    assert L.cjs_bwt_linear(d.ctypes.data, u.ctypes.data, d.size, C.byref(p)) == 0
    uo, po = oracle.bwt_linear(d)
    assert p.value == po and np.array_equal(u, uo)


def test_inverse_linear_bwt_by_list_ranking():
    """BWT.unbwtransform (lib/BWT.js:352-363) as counting-sort LF links + list ranking (K6): inverts
    the oracle's bwtransform, and agrees with the oracle's serial LF walk."""
    L = _lib.load(stagelib.EMU_SO)
    raws = [b"banana", b"a", b"ab", b"aaaa", b"mississippi", b"\0" * 50, b"ab" * 40, bytes(range(256)) * 3]
    datas = [np.frombuffer(r, dtype=np.uint8).copy() for r in raws]
    datas += [cases.case_input(c) for c in ("mary9", "text1k", "bytes40", "text100k")]
    for d in datas:
        d = np.ascontiguousarray(d)
        u, p = oracle.bwt_linear(d)
        out = np.zeros(d.size, np.uint8)
        assert L.cjs_unbwt_linear(np.ascontiguousarray(u).ctypes.data, out.ctypes.data, d.size, p) == 0
        assert np.array_equal(out, d) and np.array_equal(out, oracle.unbwt_linear(u, p))
    assert L.cjs_unbwt_linear(datas[0].ctypes.data, datas[0].ctypes.data, 0, 0) == 0          # n = 0
    out = np.zeros(6, np.uint8)
    assert L.cjs_unbwt_linear(datas[0].ctypes.data, out.ctypes.data, 6, 7) == -22              # pidx > n
    assert L.cjs_unbwt_linear(datas[0].ctypes.data, out.ctypes.data, 6, 2) == 0                # inconsistent pair: no fault
    # (T, pidx) pairs that are the BWT of nothing (several LF cycles, chains that leave [0, n)): the reference's n-step
    # walk, literally (reference-made vectors in golden.json pin the oracle on exactly these pairs)
    for cid, pidxs in cases.UNBWT_CASES.items():
        d = np.ascontiguousarray(cases.case_input(cid))
        if d.size > 2000:
            continue
        for p in pidxs:
            out = np.full(d.size, 0xEE, np.uint8)
            assert L.cjs_unbwt_linear(d.ctypes.data, out.ctypes.data, d.size, p) == 0
            assert np.array_equal(out, oracle.unbwt_linear(d, p)), (cid, p)


def test_allocator_entry_vs_reference_vectors(golden):
    """cjs_huff_lengths(_batch) = allocateHuffmanCodeLengths: the 8 KATs of test/huffman.js:15-77 and
    the 408 reference-made fuzz vectors (both branches of the length limiter)."""
    from test_oracle import HUFF_KATS
    L = _lib.load(stagelib.EMU_SO)
    for freq, maxlen, expect in HUFF_KATS:
        a = np.array(freq, dtype=np.int64)
        assert L.cjs_huff_lengths(a.ctypes.data, a.size, maxlen) == 0
        assert a.tolist() == expect
    for ml in (3, 6, 20, 32):
        cs = [c for c in golden["huff"]["cases"] if c["max_len"] == ml]
        off = np.zeros(len(cs) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(c["freq"]) for c in cs])
        flat = np.concatenate([np.array(c["freq"], dtype=np.int64) for c in cs])
        assert L.cjs_huff_lengths_batch(flat.ctypes.data, off.ctypes.data, len(cs), ml) == 0
        for k, c in enumerate(cs):
            assert flat[off[k]:off[k + 1]].tolist() == c["lengths"], (ml, k)
    a = np.ones(9, dtype=np.int64)
    assert L.cjs_huff_lengths(a.ctypes.data, 9, 3) == -22            # 9 symbols cannot fit 3-bit codes


def test_bwtc_streams_vs_reference_digest(emu_ctx, golden):
    """BWTC -6..-9 (lib/BWTC.js): linear BWT + MTF/RLE2 through the kernels (CPU debug build here),
    Fenwick model + range coder on the host; bit-identical to the reference."""
    L, h = emu_ctx
    n = 0
    for k in sorted(k for k in golden if ":bwtc:" in k):
        cid, _, lv = k.split(":")
        if cid not in ("empty", "a1", "a1000", "sample0", "text1k", "bytes40", "text100k") or k == "text100k:bwtc:7":
            continue
        d = cases.case_input(cid)
        if d is None:
            continue
        d = np.ascontiguousarray(d)
        cap = int(L.cjs_bwtc_compress_bound(d.size))
        out = np.zeros(cap, np.uint8)
        m = L.cjs_bwtc_compress(h, d.ctypes.data, d.size, int(lv), out.ctypes.data, cap, d.size)
        assert m == golden[k]["out_len"], k
        assert hashlib.sha256(out[:m].tobytes()).hexdigest() == golden[k]["out_sha256"], k
        n += 1
    assert n >= 9                                         # incl. levels 1-5 (DefSumModel, lib/BWTC.js:107)


def test_bwtc_fuzz_vs_reference_and_k10_overflow_fallback(emu_ctx):
    """A slice of the reference-made BWTC fuzz vectors (tests/golden/golden_bwtc.json; all 308 run on the GPU) through the CPU
    debug build, and the same inputs with K10's rows shrunk to nothing (CJS_K10_CAP: every block overflows and is modelled on
    the host instead): the bytes must not change."""
    import json
    import bwtc_cases
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_bwtc.json")))["vectors"]
    L, h = emu_ctx

    def run(d, lv):
        cap = int(L.cjs_bwtc_compress_bound(d.size))
        out = np.zeros(cap, np.uint8)
        m = L.cjs_bwtc_compress(h, d.ctypes.data, d.size, lv, out.ctypes.data, cap, d.size)
        assert m > 0
        return out[:m].tobytes()

    n = 0
    for i in range(0, bwtc_cases.N_SMALL, 3):
        d, lv = bwtc_cases.case(i)
        if d.size > 6000:
            continue
        v = g["fuzz%d" % i]
        o = run(d, lv)
        assert len(o) == v["out_len"] and hashlib.sha256(o).hexdigest() == v["out_sha256"], (i, lv, d.size)
        if lv >= 6 and n % 4 == 0:
            os.environ["CJS_K10_CAP"] = "64"
            try:
                assert run(d, lv) == o, (i, lv)
            finally:
                del os.environ["CJS_K10_CAP"]
        n += 1
    assert n >= 60


def test_decoder_vs_reference_vectors(emu_ctx):
    """GPU decoder (K7 entropy decode, K8 inverse BWT by splitter ranking, K9 un-RLE1 + CRC) through the
    C ABI on the CPU debug build: every small stream of the decode catalogue - valid, truncated,
    concatenated, corrupted - must do what the reference's Bzip2.decompressFile did (bytes, or the same
    Err code and detail, CRC values included)."""
    import json
    import decode_cases
    from decode_check import check_block, check_stream, check_table
    L, h = emu_ctx
    with open(os.path.join(ROOT, "tests", "golden", "golden_decode.json")) as f:
        g = json.load(f)["vectors"]
    n = 0
    by = {}
    for sid, s, ms in decode_cases.streams():
        by[sid] = s
        if s is None or len(s) > 1200:
            continue
        check_stream(L, h, sid, s, ms, g[sid])
        if g[sid]["ok"] and "table" in g[sid] and len(s) > 4:
            check_table(L, h, sid, s, ms, g[sid])
        n += 1
    assert n >= 60
    for sid, bitpos in decode_cases.BLOCK_CASES:
        if by.get(sid) is not None and len(by[sid]) <= 1200:
            check_block(L, h, sid, by[sid], bitpos, g["block:%s@%d" % (sid, bitpos)])


def test_decoder_multi_block_and_long_runs(emu_ctx):
    """Several blocks per stream (level 1), a libbzip2-made stream, and RLE1 runs that expand 50x."""
    import json
    import decode_cases
    from decode_check import check_stream
    L, h = emu_ctx
    with open(os.path.join(ROOT, "tests", "golden", "golden_decode.json")) as f:
        g = json.load(f)["vectors"]
    want = {"enc:text100k:1", "lib:text100k:9", "enc:zeros300k:1", "enc:runs300k:9", "fix:sample3", "lib:periodic_long:2"}
    for sid, s, ms in decode_cases.streams():
        if sid in want and s is not None:
            check_stream(L, h, sid, s, ms, g[sid])


def test_decoder_code_less_bits_behind_symbol_50():
    """A corrupted block whose Huffman data runs into bits that no code of the group's table matches only AFTER the group's 50th
    symbol (they belong to the next group): the wave that walks the code starts must leave them to the next group and go on.
    (It once stopped there, and the symbol wave waited for ever; found by the GPU fuzz, seed 1 case 191.)  In a subprocess with
    a timeout: a hang must fail, not block the suite."""
    stagelib.build_emu()
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import stagelib, oracle
from decode_fuzz import run_gpu
from compressjs_amd import _lib
L = _lib.load(stagelib.EMU_SO)
h = L.cjs_create(0, 2)
s = bytes.fromhex("425a683131415926535910294ff7000129118040001b69b6c0200060295514f6054dfa7a850c000e2873436a1943f50ea87543ed0d50f287743286")
want = oracle.bz2_decompress(s, False)
got = run_gpu(L, h, s, False)
assert want[0] == -5 and got[:2] == (want[0], want[1]), (got[:2], want[:2])
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.check_output([sys.executable, "-c", code], timeout=120).decode()
    assert out.strip().endswith("ok")


def test_bwtc_decode_inverts_reference_streams(emu_ctx, golden):
    """BWTC.decompressFile: host range decoder + K6 on the streams the reference's BWTC.compressFile
    produced (ours are bit-identical to them, checked above): decoding gives the input back."""
    L, h = emu_ctx
    n = 0
    for k in sorted(k for k in golden if ":bwtc:" in k):
        cid, _, lv = k.split(":")
        if cid not in ("empty", "a1", "a1000", "sample0", "text1k", "bytes40", "text100k") or k == "text100k:bwtc:7":
            continue
        d = cases.case_input(cid)
        if d is None:
            continue
        d = np.ascontiguousarray(d)
        cap = int(L.cjs_bwtc_compress_bound(d.size))
        z = np.zeros(cap, np.uint8)
        m = L.cjs_bwtc_compress(h, d.ctypes.data, d.size, int(lv), z.ctypes.data, cap, d.size)
        assert hashlib.sha256(z[:m].tobytes()).hexdigest() == golden[k]["out_sha256"], k
        out = np.zeros(max(d.size, 1), np.uint8)
        declared = C.c_int64(-7)
        r = L.cjs_bwtc_decompress(h, z.ctypes.data, m, out.ctypes.data, d.size, C.byref(declared))
        assert r == d.size and declared.value == d.size, (k, r, declared.value)
        assert np.array_equal(out[:d.size], d), k
        if d.size > 8:
            assert L.cjs_bwtc_decompress(h, z.ctypes.data, m, out.ctypes.data, d.size - 1, C.byref(declared)) == -21
            assert L.cjs_bwtc_last_size(h) == d.size
            r2 = L.cjs_bwtc_decompress(h, z.ctypes.data, m // 2, out.ctypes.data, d.size, C.byref(declared))
            assert r2 == -31 or 0 <= r2 <= d.size, (k, r2)          # truncated: an error code or a short result, never a fault
        n += 1
    assert n >= 6
    bad = np.frombuffer(b"bwtx\x81\x09", dtype=np.uint8).copy()
    assert L.cjs_bwtc_decompress(h, bad.ctypes.data, bad.size, None, 0, None) == -30            # 'Bad magic'


def test_decoder_differential_fuzz_vs_oracle(emu_ctx):
    """Random small inputs -> valid streams -> flips / truncations / insertions / concatenations: the GPU
    decoder (CPU debug build) and the oracle's decoder must agree on bytes, Err code and detail."""
    import decode_fuzz
    L, h = emu_ctx
    assert decode_fuzz.fuzz(L, h, seed=20260925, cases=150) == 150


@pytest.mark.parametrize("env_add", [{}, {"CJS_K1_CARRY": "0"}, {"CJS_TEXT_BYTES": "0"}, {"CJS_TEXT_BYTES": "44", "CJS_DEEP_LANE_CAP": "0"},
                                     {"CJS_BSORT_ITERS": "0", "CJS_DEEP_BIG_DIV": "1"}, {"CJS_BSORT_ITERS": "2"},
                                     {"CJS_DEEP_BIG_DIV": "1000000000", "CJS_K1_SYNC": "0"}])
def test_deep_refinement_of_suffix_sort(env_add):
    """The text stages (in-bucket iterations, list-driven refinement rounds, lane kernels) resolve groups by comparing the
    text before any rank exists; what they leave (long repeats, identical rotations, groups that stay big) goes to the
    doubling rounds, which are skipped when nothing is left.  The knobs of k1_bwt.hip (k1_knobs) against the oracle, on inputs
    that end in each continuation: text stages off, a short cap with no lane kernels, no / two in-bucket iterations, the
    predictor forcing the text stages on / off, no read-back."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib
from compressjs_amd import synth
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic.restype = C.c_int32
L.cjs_bwt_cyclic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
passage = synth.text_like(700, 3)
cases = [np.concatenate([synth.text_like(9000, 1), passage, synth.text_like(5000, 2), passage, synth.runs_mixed(3000, 1), passage]),
         synth.enwik_like(40000, 7), synth.lcg_ascii(5000, 3), synth.runs_mixed(12000, 5), synth.periodic(6000),
         np.tile(synth.text_like(1000, 5), 9), np.zeros(3000, np.uint8), synth.text_like(65, 2), synth.periodic(70),
         np.tile(synth.text_like(300, 8), 30)]
res = []
for d in cases:
    d = np.ascontiguousarray(d)
    u = np.zeros(d.size, np.uint8); p = C.c_uint32(0)
    assert L.cjs_bwt_cyclic(d.ctypes.data, u.ctypes.data, d.size, C.byref(p)) == 0
    uo, po = oracle.bwt_cyclic(d)
    assert p.value == po and (u == uo).all(), d.size
    res.append((L.cjs_dbg_k1_rounds(), L.cjs_dbg_k1_sparse_rounds()))
print("ok", res)
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    env = dict(os.environ, CJS_K1_TRACE="1", **env_add)                  # (the trace reads the rounds' list counters back)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=900, stderr=subprocess.DEVNULL).decode()
    assert out.startswith("ok")
    if not env_add:
        rounds = eval(out[2:])
        assert rounds[1] == (0, 0) and rounds[2] == (0, 0), rounds      # phrase-reuse text, random: no doubling round at all
        assert rounds[0][1] >= 1, rounds                                  # 700-byte repeats: left to the doubling rounds


def _bwt_batch(L, blocks):
    cap = max(b.size for b in blocks)
    T = np.zeros(len(blocks) * cap, np.uint8)
    nl = np.zeros(len(blocks), np.uint32)
    for i, b in enumerate(blocks):
        T[i * cap:i * cap + b.size] = b
        nl[i] = b.size
    U = np.zeros_like(T)
    pidx = np.zeros(len(blocks), np.uint32)
    L.cjs_bwt_cyclic_batch.restype = C.c_int32
    L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, len(blocks), cap, U.ctypes.data, pidx.ctypes.data) == 0
    return [(U[i * cap:i * cap + b.size], int(pidx[i])) for i, b in enumerate(blocks)]


def _front_blocks():
    rng = np.random.RandomState(7)
    heavy = synth.text_like(60000, 9).copy()                 # one 8-byte key in ~1 % of the positions, not enough quantiles
    for p in rng.randint(0, heavy.size - 8, size=500):
        heavy[p:p + 8] = np.frombuffer(b"QQQQZZZZ", np.uint8)
    # heavy keys at several depths (the task levels of k1_front.hip: one-key buckets beyond LDS, groups above 256 rotations):
    # a 40-byte phrase at 3000 places, and 2500 "table rows" that share 60 bytes before and 30 bytes after a varying field
    deep = synth.text_like(400000, 13).copy()
    phrase = np.frombuffer(b"<td class=\"cell numeric\" align=\"right\">", np.uint8)
    for p in rng.randint(0, deep.size - 64, size=3000):
        deep[p:p + phrase.size] = phrase
    rows = np.concatenate([np.frombuffer(b"<tr><td class=\"c1\">row</td><td class=\"c2\" style=\"width:10px\">" + (b"%05d" % int(v)) +
                                         b"</td><td>fixed tail of the row</td></tr>\n", np.uint8) for v in rng.randint(0, 100000, size=2500)])
    return [deep, rows, synth.text_like(50000, 11), synth.enwik_like(40000, 12), synth.lcg_ascii(30000, 3),
            synth.periodic(25001, b"ab"), synth.periodic(20011, b"the quick brown fox jumps over the lazy dog\n"),
            np.zeros(12000, np.uint8), synth.runs_mixed(40000, 4), heavy,
            rng.randint(0, 256, size=23000).astype(np.uint8), rng.randint(97, 99, size=15000).astype(np.uint8),
            np.full(5000, 255, np.uint8), synth.text_like(4097, 1), synth.text_like(4096, 2)]


@pytest.mark.parametrize("variant", ["default", "tiny_buckets", "doubling_forced"])
def test_sample_sort_front_end(variant):
    """k1_front.hip (sample-sort front end of the suffix sort) on blocks large enough to be partitioned: text,
    random, periodic (pure buckets), runs, a moderately heavy key; `tiny_buckets` is a build with a 256-rotation
    bucket capacity and 2 samples per bucket, so that the oversize path (task levels) runs all the time - the path blocks
    above ~1.1 MB take as a matter of course (test_bwt_entry_points_above_one_megabyte).  BWT + origPtr of every block
    against the oracle."""
    env = dict(os.environ)
    if variant == "doubling_forced":
        # round 6: the predictor forced on (no text stages) with 16-byte keys - heavy keys spread over sub-buckets by their own samples' sub-splitters,
        # buckets of one 16-byte key left as groups, task levels in front of doubling rounds that start at h = 16
        env["CJS_DEEP_BIG_DIV"] = "1073741824"
        variant = "default"
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'));"
            "sys.path.insert(0, os.path.join(%r, 'tests', 'golden'));"
            "import test_emu_pipeline as t; t._front_check(%r)" % (ROOT, ROOT, ROOT, variant))
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout + r.stderr


def _front_check(variant):
    so = stagelib.build_emu()
    if variant == "tiny_buckets":
        so = os.path.join(ROOT, "tests", "emu", "libcjs_emu_tiny.so")
        import subprocess
        subprocess.check_call(["sh", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL,
                              env=dict(os.environ, EMU_OUT=so, EMU_DEFS="-DK1F_C=256 -DK1F_OVS=2"))
    L = C.CDLL(so)
    blocks = _front_blocks()
    for (u, p), b in zip(_bwt_batch(L, blocks), blocks):
        uo, po = oracle.bwt_cyclic(b)
        assert p == po and np.array_equal(u, uo), (variant, b.size, bytes(b[:16]))
    # linear mode (BWT.bwtransform: suffixes, no wrap-around) through the same front end: heavy keys must not be keyed deeper
    L.cjs_bwt_linear.restype = C.c_int32
    L.cjs_bwt_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    for b in blocks[:2] + blocks[4:6]:
        u = np.zeros(b.size, np.uint8)
        pi = C.c_uint32(0)
        assert L.cjs_bwt_linear(b.ctypes.data, u.ctypes.data, b.size, C.byref(pi)) == 0
        uo, po = oracle.bwt_linear(b)
        assert pi.value == po and np.array_equal(u, uo), (variant, "linear", b.size)


def test_doubling_rounds_variants():
    """k1_dbl.hip's alternative code paths against the oracle: `packed_off` is a build in which no block is small enough for the
    packed-word paths (K1D_PACK_MAXN = K1D_RADIX_MAXN = 0: k1d_round counts on plain keys, k1d_med sorts (key, rotation) pairs by
    the bitonic network - what blocks of 2^20 bytes and more take) with the S-group threshold at 64; both builds run with the text
    stages off (CJS_TEXT_BYTES=0: every tie goes through the doubling rounds, from 8 bytes) on text, runs, periodic and tiled
    inputs, groups of every size class, cyclic and linear.  CJS_K1_PERIOD=0: the periodic blocks go through the rounds too (their
    closed form has its own test below)."""
    import subprocess
    import sys
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'));"
            "sys.path.insert(0, os.path.join(%r, 'tests', 'golden'));"
            "import test_emu_pipeline as t; t._doubling_check(sys.argv[1])" % (ROOT, ROOT, ROOT))
    for variant in ("default", "packed_off"):
        r = subprocess.run([sys.executable, "-c", code, variant], env=dict(os.environ, CJS_TEXT_BYTES="0", CJS_K1_PERIOD="0"), capture_output=True, text=True, timeout=3000)
        assert r.returncode == 0, variant + r.stdout + r.stderr


def _doubling_check(variant):
    so = stagelib.build_emu()
    if variant == "packed_off":
        so = os.path.join(ROOT, "tests", "emu", "libcjs_emu_dbl.so")
        import subprocess
        subprocess.check_call(["sh", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL,
                              env=dict(os.environ, EMU_OUT=so, EMU_DEFS="-DK1D_PACK_MAXN=0u -DK1D_RADIX_MAXN=0u -DK1D_GS=64u"))
    L = C.CDLL(so)
    for f in (L.cjs_bwt_cyclic, L.cjs_bwt_linear):
        f.restype = C.c_int32
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    rng = np.random.RandomState(3)
    blocks = [synth.text_like(60_000, 3), synth.enwik_like(50_000, 4), synth.runs_mixed(40_000, 5),
              synth.periodic(20_000, b"ab"), synth.periodic(44 * 500, b"the quick brown fox jumps over the lazy dog\n"),
              synth.periodic(30_001, b"the quick brown fox jumps over the lazy dog\n"), np.tile(synth.text_like(3000, 5), 8),
              np.full(30_000, 7, np.uint8), rng.randint(97, 99, size=40_000).astype(np.uint8),
              np.tile(np.concatenate([np.full(9000, 1, np.uint8), np.full(9000, 2, np.uint8)]), 2)]
    for b in blocks:
        b = np.ascontiguousarray(b)
        for f, o in ((L.cjs_bwt_cyclic, oracle.bwt_cyclic), (L.cjs_bwt_linear, oracle.bwt_linear)):
            u = np.zeros(b.size, np.uint8)
            p = C.c_uint32(0)
            assert f(b.ctypes.data, u.ctypes.data, b.size, C.byref(p)) == 0
            uo, po = o(b)
            assert p.value == po and np.array_equal(u, uo), (variant, b.size, bytes(b[:12]))


def test_periodic_blocks_closed_form():
    """k1_period.hip: blocks with a linear period p <= 64 get their suffix array from the closed form (phase order, one sign
    for the order inside a phase, the p - 1 rotations that start in the last period placed by binary search).  Every p in 1..64
    with block lengths that are and are not multiples of p, alphabets of 2..256 symbols (non-primitive period words included:
    the detector must find the smallest period), periods beyond 64 and near-periodic blocks (one defect: early, late, last
    byte), which must fall through to the general sort; all against the oracle.  CJS_K1_TRACE counts the blocks taken."""
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(11)
cap = 8192
def run(blocks):
    nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for i, d in enumerate(blocks):
        T[i, :d.size] = d; nl[i] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    for i, d in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[i] == po and (U[i, :d.size] == uo).all(), (d.size, bytes(d[:70]))
    return L.cjs_dbg_k1_periodic_blocks()
def mk(p, n, alpha):
    return np.tile(rng.integers(0, alpha, p).astype(np.uint8), n // p + 2)[:n].copy()
per, oth = [], []
for p in range(1, 65):
    for k in range(2):
        n = int(rng.integers(4096, 8000))
        if k == 1:
            n = (n // p) * p
        per.append(mk(p, n, int(rng.choice([2, 3, 4, 256]))))
per.append(np.frombuffer(b"\0\0\0\0\xfb" * 1000 + b"\0\0\0", np.uint8).copy())      # zeros after RLE1
per.append(np.frombuffer(b"ab" * 2500 + b"a", np.uint8).copy())
for p in (65, 70, 100, 1000):
    q = rng.integers(0, 256, p).astype(np.uint8); q[0] = 255; q[1:] %%= 255
    oth.append(np.tile(q, 6000 // p + 2)[:6000].copy())
for at in (5999, 3000, 2047, 100):
    d = mk(7, 6000, 3); d[at] ^= 1; oth.append(d)
oth.append(mk(3, 4000, 2))                                  # too short for the closed form
seen = 0
for i in range(0, len(per), 16):
    seen += run(per[i:i + 16])
assert seen == len(per), (seen, len(per))
assert run(oth) == 0
print("ok", seen)
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CJS_K1_TRACE="1"), capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]


@pytest.mark.parametrize("env_add", [{}, {"CJS_DEEP_BIG_DIV": "1073741824"}], ids=["default", "doubling_forced"])
def test_long_period_words_same_phase_groups(env_add):
    """k1d_round's shortcut for blocks reduced to three periods (a group whose members share s mod p is settled by index order with the
    block's sign): tests/period_stress.py - period words that are text, binary noise, Q^m with defects (groups of several phases), one
    phrase planted many times, 0xFF-heavy, and tests/periodwords.py's families; p from 65 to n / 4, n = 0, 1, p - 1 or anything (mod p) -
    16 blocks per setting against the oracle's cyclic BWT; with the doubling path forced every triple of the reduced block goes through it."""
    import subprocess
    import sys
    stagelib.build_emu()
    r = subprocess.run([sys.executable, os.path.join(stagelib.ROOT, "tests", "period_stress.py"), "31", "16"],
                       env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and r.stdout.startswith("ok 16"), r.stdout + r.stderr[-3000:]


def test_long_period_blocks_sorted_through_three_periods():
    """k1_period.hip, periods beyond 64: a block T of period p (n >= 16384, 3 p + n mod p <= 0.8 n) is sorted as its first
    n' = 3 p + (n mod p) bytes (n' = p when p divides n) and expanded - every phase's missing members next to the first one the
    reduced block has, in the phase's index order.  Period words of four kinds (binary, text, random bytes, half zeros: the last
    and files of equal lines tiled need the second detection stage), p from 65 to n / 4, multiples of p and not; transform AND origPtr against
    the oracle (the order inside a phase only shows in origPtr and in the bytes in front of rotation 0)."""
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib
from compressjs_amd import synth
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(1)
cap = 40000
def run(blocks):
    nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for i, d in enumerate(blocks):
        T[i, :d.size] = d; nl[i] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    for i, d in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[i] == po and (U[i, :d.size] == uo).all(), (d.size, int(P[i]), po)
    return L.cjs_dbg_k1_periodic_blocks() >> 16
blocks = []
for it in range(32):
    n = int(rng.integers(16384, 39000))
    p = int(rng.integers(65, n // 4))
    kind = it %% 4
    if kind == 0: P = rng.integers(0, 2, p).astype(np.uint8)
    elif kind == 1: P = synth.text_like(p, it)
    elif kind == 2: P = rng.integers(0, 256, p).astype(np.uint8)
    else: P = np.concatenate([np.zeros(p // 2, np.uint8), rng.integers(0, 3, p - p // 2).astype(np.uint8)])
    if it %% 5 == 0: n = (n // p) * p
    blocks.append(np.tile(P, n // p + 2)[:n].copy())
for fl, n, ll in ((3000, 30001, 81), (4000, 36000, 81), (3001, 30001, 4), (2999, 38000, 1)):   # a file of equal lines (cut in a line), tiled: found by the second stage
    f = np.concatenate([np.tile(synth.text_like(ll, 3), fl // ll + 1)[:fl - 9], np.frombuffer(b"the end.\n", np.uint8)])
    blocks.append(np.tile(f, n // fl + 1)[:n].copy())
d = np.tile(synth.text_like(3000, 9), 10)[:29000].copy(); d[-1] ^= 1; blocks.append(d)       # one foreign byte: the general sort
d = np.tile(synth.text_like(9000, 9), 4)[:30000].copy(); blocks.append(d)                    # 3 p + r0 > 0.8 n: not worth reducing
seen = 0
for i in range(0, len(blocks), 8):
    seen += run(blocks[i:i + 8])
assert 26 <= seen <= 36, seen
print("ok", seen)
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CJS_K1_TRACE="1"), capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]


def test_adversarial_period_words():
    """k1_period.hip against period words chosen to break it (tests/periodwords.py: Fibonacci and Thue-Morse prefixes, (w)^m with one
    defect, a^k b, long borders u v u, a short period with one flipped byte, tripled runs, binary noise; p in 2..64, 65..200, just
    below n / 4 and in between; n = 0, 1, p - 1 mod p): 120 blocks of 16 384 .. 36 000 bytes, transform and origPtr against the oracle;
    both routes (closed form, three-period reduction) must have been taken.  (1 200 such cases, no mismatch - pinned here.)"""
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib, periodwords
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(77)
cap = 36000
B = [b for b in periodwords.blocks(cap, rng, 160, small=True) if b[1].size >= 16384 or int(b[0].split("p=")[1].split()[0]) <= 64][:120]
closed = red = 0
for i in range(0, len(B), 8):
    blocks = B[i:i + 8]
    nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for j, (_, d) in enumerate(blocks):
        T[j, :d.size] = d; nl[j] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    r = L.cjs_dbg_k1_periodic_blocks()
    closed += r & 0xFFFF; red += r >> 16
    for j, (name, d) in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[j] == po and (U[j, :d.size] == uo).all(), name
assert closed >= 10 and red >= 30, (closed, red)
print("ok", len(B), closed, red)
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CJS_K1_TRACE="1"), capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]


def test_segmented_host_pipeline():
    """cjs_bz2_compress on inputs longer than 1.5 segments: planned and encoded segment by segment (upload / encode /
    download overlapped by two helper threads).  CJS_SEG_BYTES=120000 makes a segment ~1.2 level-1 blocks, so the
    "drop the last block and restart there" seam, the segment-doubling path (runs: one block swallows a segment) and the
    incremental download all run; the stream must equal the oracle's."""
    import subprocess
    import sys
    code = ("import sys, os; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'));"
            "sys.path.insert(0, os.path.join(%r, 'tests', 'golden'));"
            "import test_emu_pipeline as t; t._segmented_check()" % (ROOT, ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CJS_SEG_BYTES="120000"), capture_output=True,
                       text=True, timeout=3000)
    assert r.returncode == 0, r.stdout + r.stderr


def _segmented_check():
    L = _lib.load(stagelib.build_emu())
    h = L.cjs_create(0, 4)
    try:
        for d, lv in ((synth.text_like(330_000, 21), 1), (synth.runs_mixed(300_000, 6), 1), (synth.text_like(185_000, 2), 1)):
            assert _compress((L, h), d, lv) == oracle.bz2_compress(d, lv), (d.size, lv)
        # the same inputs over three contexts (cjs_bz2_compress_multi: segment k on context k mod 3, windows, bit-shifted
        # placement, seam bytes, trailer); zeros / runs take the replicated plan (every context plans the whole input, encodes its share)
        hs = [L.cjs_create(0, 4) for _ in range(3)]
        arr = (C.c_void_p * 3)(*hs)
        runs = synth.text_like(520_000, 8).copy()          # five segments on three contexts (two waves), runs on / across the cuts,
        runs[119_990:120_004] = 65                          # a block boundary (99 981) right where a straddling run starts: refuse or agree
        runs[239_998:240_001] = 66
        runs[359_000:361_500] = 67
        took = []
        L.cjs_dbg_multi_fallbacks.restype = C.c_int
        for d, lv in ((np.zeros(260_000, np.uint8), 1), (synth.enwik_like(330_000, 4), 1), (synth.text_like(520_000, 7), 1), (runs, 1),
                      (np.concatenate([synth.lcg_ascii(99_981, 3), np.full(20_030, 65, np.uint8), synth.lcg_ascii(140_000, 4)]), 1)):
            cap = int(L.cjs_bz2_compress_bound(d.size))
            out = np.full(cap, 0xAA, np.uint8)                  # stale bytes: the call must write every byte it returns
            fb = L.cjs_dbg_multi_fallbacks()
            n = L.cjs_bz2_compress_multi(arr, 3, d.ctypes.data, d.size, lv, out.ctypes.data, cap)
            assert n > 0 and out[:n].tobytes() == oracle.bz2_compress(d, lv), (d.size, lv, n)
            took.append(L.cjs_dbg_multi_fallbacks() - fb)
        assert took[0] == 1 and took[1] == 0 and took[2] == 0 and sum(took) >= 2, took     # zeros: replicated; the two text streams: the parallel plan
        # round 6: a block boundary inside a run of 4..9 equal bytes (ordinary text has them) is CARRIED through the segments' chain - the
        # segments behind it are planned again from the moved target - instead of sending the call to the replicated plan
        import torch
        from compressjs_amd.bzip2 import Context
        L.cjs_dbg_multi_replans.restype = C.c_int
        carried = synth.text_like(520_000, 7).copy()
        saved = _lib._lib
        _lib._lib = L
        try:
            pc = Context(0, 2)
            nser = pc.plan(torch.from_numpy(carried), 1)
            bnd = [pc.plan_block_start(k) for k in range(nser)]
            pc.close()
        finally:
            _lib._lib = saved
        carried[bnd[1] - 3:bnd[1] + 3] = 66                 # boundary 1 (segment 0) inside a run of 6; boundary 3 inside a run of 9
        carried[bnd[3] - 2:bnd[3] + 7] = 67
        cap = int(L.cjs_bz2_compress_bound(carried.size))
        out = np.full(cap, 0xAA, np.uint8)
        fb, rp = L.cjs_dbg_multi_fallbacks(), L.cjs_dbg_multi_replans()
        n = L.cjs_bz2_compress_multi(arr, 3, carried.ctypes.data, carried.size, 1, out.ctypes.data, cap)
        assert n > 0 and out[:n].tobytes() == oracle.bz2_compress(carried, 1)
        assert L.cjs_dbg_multi_fallbacks() == fb and L.cjs_dbg_multi_replans() > rp, (L.cjs_dbg_multi_fallbacks() - fb, L.cjs_dbg_multi_replans() - rp)
        for x in hs:
            L.cjs_destroy(x)
    finally:
        L.cjs_destroy(h)


@pytest.mark.parametrize("env_add", [{}, {"CJS_TEXT_BYTES": "0"}, {"CJS_DEEP_BIG_DIV": "1073741824"}],
                         ids=["default", "text_stages_off", "doubling_path_from_16"])
def test_attack_words_on_the_bucket_sort(env_add):
    """tests/attackwords.py (the families the round-5 judge attacked the 16-byte-key bucket sort with: words of 15 / 16 / 17 and 40..90
    bytes over {0xFE,0xFF}, {a..d} and the full alphabet, 0xFF- / 0x00-heavy blocks with sparse defects, one phrase of 16..64 bytes in
    300..1500 places, near-periodic words with flipped bits, a mixed block): 28 blocks of 8 000 .. 48 000 bytes per knob setting (the GPU
    suite runs the same families at the -9 block capacity, also with CJS_K1_CARRY=0),
    transform and origPtr against the oracle."""
    import subprocess
    import sys
    code = r'''
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle, stagelib, attackwords
from compressjs_amd import synth
L = C.CDLL(stagelib.build_emu())
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(20260930)
cap = 48000
B = attackwords.blocks(cap, rng, synth.text_like, small=True)
for i in range(0, len(B), 8):
    blocks = B[i:i + 8]
    nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for j, (_, d) in enumerate(blocks):
        T[j, :d.size] = d; nl[j] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    for j, (name, d) in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[j] == po and (U[j, :d.size] == uo).all(), name
print("ok", len(B))
''' % (stagelib.ROOT, os.path.join(stagelib.ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]
