"""Parity tests proper: the HIP path (through the C ABI) against the oracle, the reference-made
golden digests, and size-independent properties at BASELINE.json's full size.  Run with -m gpu."""
import bz2
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import cases
import oracle
import stagelib
import workloads
from compressjs_amd import BWT, Bzip2, _lib, synth
from compressjs_amd.bzip2 import Context

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def ctx():
    c = Context(0, 128)
    yield c
    c.close()


def test_every_golden_stream_digest(golden):
    """Bit-identical .bz2 output to the reference (node 12) on EVERY pinned input, incl. test/sample0..5.ref at
    -1 and -9.  The reference's fixtures travel to the GPU box under oracle/_ref/fixtures
    (__graft_entry__.build() stages them); without them this test fails, it does not skip."""
    keys = sorted(k for k in golden if k.split(":")[1:2] == ["bz2"])
    assert len(keys) == sum(len(lv) for _, lv in cases.CASES.values()), "golden.json out of date: rerun tests/golden/make_golden.py"
    missing = sorted({k.split(":")[0] for k in keys if cases.case_input(k.split(":")[0]) is None})
    assert not missing, "reference fixtures not staged for %s (run __graft_entry__.build() in the build container)" % missing
    n = 0
    for k in keys:
        cid, _, lv = k.split(":")
        o = Bzip2.compressFile(cases.case_input(cid), None, int(lv))
        assert len(o) == golden[k]["out_len"], k
        assert _sha(o) == golden[k]["out_sha256"], k
        n += 1
    assert n == len(keys) and n >= 49


def test_sample5_level9_headline_digest(golden):
    d = cases.case_input("sample5")
    if d is None:
        pytest.skip("test/sample5.ref not staged (run __graft_entry__.build() in the build container)")
    o = Bzip2.compressFile(d, None, 9)
    assert len(o) == 274768 and _sha(o).startswith("236be53bab8972f04032ef88")


def test_stage_by_stage_vs_oracle_full_blocks():
    L = stagelib.load("real")
    d = np.concatenate([synth.text_like(1_300_000, 31), synth.runs_mixed(400_000, 8), synth.lcg_ascii(500_000, 9)])
    orc = list(oracle.block_stages(d, 9))
    dev = stagelib.block_stages(L, [o["T"] for o in orc], 899981, 5, crcs=[o["crc"] for o in orc])
    for dv, o in zip(dev, orc):
        assert stagelib.compare_with_oracle(dv, o, 5) == []
    assert dev[0]["stream"] == oracle.bz2_compress(d, 9)


def test_all_levels_small_and_medium():
    d = synth.text_like(1_050_000, 77)
    for lv in range(1, 10):
        assert Bzip2.compressFile(d, None, lv) == oracle.bz2_compress(d, lv), lv


def test_bwt_kats_and_properties():
    for inp, out, idx in [(b"bcababa", b"cbbaaab", 5), (b"banana", b"nnbaaa", 3), (b"aaaa", b"aaaa", 3)]:
        U = np.zeros(len(inp), np.uint8)
        assert BWT.bwtransform2(np.frombuffer(inp, np.uint8), U, len(inp)) == idx
        assert U.tobytes() == out
    t = synth.text_like(899981, 3)
    U = np.zeros(t.size, np.uint8)
    p = BWT.bwtransform2(t, U, t.size)
    uo, po = oracle.bwt_cyclic(t)
    assert p == po and np.array_equal(U, uo)
    assert np.array_equal(np.sort(U), np.sort(t))           # a permutation of the block


def test_bench_stream_whole_vs_oracle_and_reference_digest(ctx, golden_big):
    """The bench workload itself (workloads.stream('enwik', 10^8), BASELINE.json configs[2]): ALL 112 blocks equal the
    oracle's stream byte for byte, the sha256 equals what the reference itself produced under node 12
    (golden_big.json), plus run-to-run and batch-size invariance and an independent decoder."""
    d = workloads.stream("enwik", 100_000_000)
    g = golden_big["enwik:100000000:bz2:9"]
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["in_sha256"], "generator drifted from the bytes the reference was run on"
    a = ctx.compress(d, 9)
    assert len(a) == g["out_len"] and _sha(a) == g["out_sha256"]
    assert a == oracle.bz2_compress(d, 9)                 # every block, not a prefix (about 25 s of one host core)
    assert ctx.compress(d, 9) == a
    small = Context(0, 7)
    try:
        assert small.compress(d, 9) == a
    finally:
        small.close()
    assert bz2.decompress(a) == d.tobytes()


@pytest.mark.parametrize("name", ["e8sa", "lcg", "e8sb"])
def test_full_size_workloads_vs_reference_digest(ctx, golden_big, name):
    """The other 10^8-byte streams of SURVEY.md 8(d) - E8S-A (test/sample5.ref || sample4.ref tiled), cfg4's random
    ASCII per GPU, E8S-B - against the digest of the reference's own output, and the leading blocks against the oracle."""
    if name in ("e8sa", "e8sb"):
        assert workloads.have_fixtures(), "reference fixtures not staged (run __graft_entry__.build() in the build container)"
    d = workloads.stream(name, 100_000_000)
    g = golden_big["%s:100000000:bz2:9" % name]
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["in_sha256"]
    a = ctx.compress(d, 9)
    assert len(a) == g["out_len"] and _sha(a) == g["out_sha256"]
    head = d[:5_000_000]
    ref = oracle.bz2_compress(head, 9)
    nbytes = (32 + sum(b["bit_len"] for b in list(oracle.block_stages(head, 9))[:4])) // 8
    assert a[:nbytes] == ref[:nbytes]


def test_random_ascii_stream(ctx):
    d = synth.lcg_ascii(20_000_000, 7)                     # BASELINE.json configs[3] shape
    a = ctx.compress(d, 9)
    assert bz2.decompress(a) == d.tobytes()
    assert a[:1000] == oracle.bz2_compress(d[:2_000_000], 9)[:1000]


def test_sharded_path_world1_equals_plain(ctx):
    import torch
    from compressjs_amd.dist import sharded_compress
    d = synth.text_like(5_000_000, 8)
    t = torch.from_numpy(d.copy()).cuda()
    out = sharded_compress(ctx, t, 9)
    assert out.cpu().numpy().tobytes() == ctx.compress(d, 9)


def test_stream_input_and_output_coercions():
    class In:
        def __init__(self, b):
            self.b, self.i = b, 0

        def readByte(self):
            if self.i >= len(self.b):
                return -1
            self.i += 1
            return self.b[self.i - 1]

    class Out:
        def __init__(self):
            self.buf = bytearray()

        def writeByte(self, b):
            self.buf.append(b)

    data = b"hello hello hello hello"
    ref = oracle.bz2_compress(data, 9)
    assert Bzip2.compressFile(In(data)) == ref
    o = Out()
    assert Bzip2.compressFile(data, o) is o and bytes(o.buf) == ref
    assert Bzip2.compressFile(list(data), len(ref)) == ref
    with pytest.raises(TypeError):
        Bzip2.compressFile(data, len(ref) + 1)
    assert Bzip2.decompressFile(ref) == data


def test_bwtc_and_linear_bwt_goldens(golden, ctx):
    """BASELINE.json configs[4] path: BWTC -6..-9 streams and BWT.bwtransform / suffixsort vectors."""
    n = 0
    for k in sorted(k for k in golden if ":bwtc:" in k):
        cid, _, lv = k.split(":")
        d = cases.case_input(cid)
        if d is None:
            continue
        o = ctx.bwtc_compress(d, int(lv))
        assert len(o) == golden[k]["out_len"] and _sha(o) == golden[k]["out_sha256"], k
        n += 1
    assert n >= 22
    for k in sorted(k for k in golden if k.endswith(":bwt")):
        cid = k.split(":")[0]
        d = cases.case_input(cid)
        if d is None:
            continue
        U = np.zeros(max(d.size, 1), np.uint8)
        p = BWT.bwtransform(d, U, None, d.size)
        assert p == golden[k]["pidx"] and _sha(U[:d.size]) == golden[k]["u_sha256"], k
        SA = np.zeros(max(d.size, 1), np.int32)
        BWT.suffixsort(d, SA, d.size)
        assert _sha(SA[:d.size].astype("<i4").tobytes()) == golden[cid + ":sa"]["sa_sha256"], k
    from compressjs_amd import BWTC
    for cid, lv in (("text100k", 9), ("bytes40", 6), ("runs300k", 8), ("text2500k", 8), ("empty", 9), ("a1", 9),
                    ("text950k", 5), ("lcg250000", 4), ("zeros300k", 1)):           # 1-5: DefSumModel
        d = cases.case_input(cid)
        z = ctx.bwtc_compress(d, lv)
        assert _sha(z) == golden["%s:bwtc:%d" % (cid, lv)]["out_sha256"]
        assert BWTC.decompressFile(z) == d.tobytes(), cid            # BWTC.decompressFile: host range decoder + K6
    with pytest.raises(RuntimeError, match="Bad magic"):
        BWTC.decompressFile(b"bwtx\x81\x09")
    big = synth.text_like(899_000, 41)
    U = np.zeros(big.size, np.uint8)
    p = BWT.bwtransform(big, U, None, big.size)
    uo, po = oracle.bwt_linear(big)
    assert p == po and np.array_equal(U, uo)
    assert np.array_equal(oracle.unbwt_linear(U, p), big)         # BWT.unbwtransform inverts it
    back = np.zeros(big.size, np.uint8)
    BWT.unbwtransform(U, back, None, big.size, p)                 # K6: list ranking on the GPU
    assert np.array_equal(back, big)
    for raw in (b"banana", b"a", b"\0" * 70000, bytes(range(256)) * 20):
        d = np.frombuffer(raw, dtype=np.uint8).copy()
        u, pp = oracle.bwt_linear(d)
        back = np.zeros(d.size, np.uint8)
        BWT.unbwtransform(u, back, None, d.size, pp)
        assert np.array_equal(back, d)
    # (T, pidx) pairs no BWT produces: same bytes as the reference's n-step walk (reference-made digests)
    nun = 0
    for k in [k for k in golden if k.split(":")[1:2] == ["unbwt"]]:
        cid, _, pidx = k.split(":")
        d = np.ascontiguousarray(cases.case_input(cid))
        back = np.full(d.size, 0xEE, np.uint8)
        BWT.unbwtransform(d, back, None, d.size, int(pidx))
        assert _sha(back) == golden[k]["u_sha256"], k
        nun += 1
    assert nun == sum(len(v) for v in cases.UNBWT_CASES.values())


def test_allocator_entry_kats_and_fuzz(golden):
    """allocateHuffmanCodeLengths through the C ABI on the GPU: test/huffman.js KATs + reference-made fuzz."""
    from compressjs_amd import HuffmanAllocator
    from test_oracle import HUFF_KATS
    for freq, maxlen, expect in HUFF_KATS:
        a = list(freq)
        HuffmanAllocator.allocateHuffmanCodeLengths(a, maxlen)
        assert a == expect
    for ml in (3, 6, 20, 32):
        cs = [c for c in golden["huff"]["cases"] if c["max_len"] == ml]
        got = HuffmanAllocator.allocate_many([c["freq"] for c in cs], ml)
        assert got == [c["lengths"] for c in cs]


def test_error_codes_through_the_abi(ctx):
    d = synth.text_like(200_000, 1)
    out = np.zeros(1000, np.uint8)
    rc = ctx.L.cjs_bz2_compress(ctx.h, d.ctypes.data, d.size, 9, out.ctypes.data, out.size)
    assert rc == -21                                   # CJS_E_NOSPACE
    assert ctx.L.cjs_bz2_compress(ctx.h, d.ctypes.data, d.size, 0, out.ctypes.data, out.size) == -20
    with pytest.raises(ValueError, match="Invalid block size multiplier"):
        Bzip2.compressFile(d, None, 11)
    # the context is still usable afterwards
    assert ctx.compress(d, 9) == oracle.bz2_compress(d, 9)


def test_level1_many_batches():
    """level 1: 99 981-byte blocks, more blocks than one batch holds (batches of 16)."""
    c = Context(0, 16)
    try:
        d = np.concatenate([synth.text_like(3_000_000, 9), synth.runs_mixed(1_500_000, 2), synth.lcg_ascii(700_000, 5)])
        a = c.compress(d, 1)
        assert a == oracle.bz2_compress(d, 1)
    finally:
        c.close()


def test_decoder_every_reference_vector(ctx):
    """The GPU decoder against all 138 reference-made decode outcomes (tests/golden/golden_decode.json):
    valid streams of both encoders, truncations, concatenations with/without multistream, bit flips."""
    import json
    import os
    import decode_cases
    from decode_check import check_block, check_stream, check_table
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_decode.json")) as f:
        g = json.load(f)["vectors"]
    L, h = ctx.L, ctx.h
    by, n = {}, 0
    for sid, s, ms in decode_cases.streams():
        by[sid] = s
        if s is None:
            continue
        check_stream(L, h, sid, s, ms, g[sid])
        if g[sid]["ok"] and "table" in g[sid] and len(s) > 4:
            check_table(L, h, sid, s, ms, g[sid])
        n += 1
    assert n >= 120            # 133 when test/sample*.bz2 are staged
    for sid, bitpos in decode_cases.BLOCK_CASES:
        if by.get(sid) is not None:
            check_block(L, h, sid, by[sid], bitpos, g["block:%s@%d" % (sid, bitpos)])


def test_decoder_round_trip_full_size(ctx):
    """compress -> decompress on the GPU at BASELINE sizes (10^7 B text, -9 and -1), plus libbzip2's
    encoder output for the same bytes; API-level wrappers."""
    d = synth.text_like(10_000_000, 77)
    raw = d.tobytes()
    for lv in (9, 1):
        z = ctx.compress(d, lv)
        assert ctx.decompress(np.frombuffer(z, dtype=np.uint8)) == raw
    z = bz2.compress(raw, 5)
    assert Bzip2.decompressFile(z) == raw
    tab = ctx.table(np.frombuffer(z, dtype=np.uint8))
    assert sum(sz for _, sz in tab) == len(raw) and tab[0][0] == 32
    first = Bzip2.decompressBlock(z, tab[1][0])
    assert first == raw[tab[0][1]:tab[0][1] + tab[1][1]]
    with pytest.raises(TypeError) as ei:
        Bzip2.decompressFile(z[:len(z) // 2])
    assert ei.value.errorCode == -5
    import torch
    zi = torch.frombuffer(bytearray(z), dtype=torch.uint8).cuda()
    out = torch.empty(len(raw) + 16, dtype=torch.uint8, device="cuda")
    assert ctx.decompress_device(zi, out) == len(raw)
    assert out[:len(raw)].cpu().numpy().tobytes() == raw


def test_decoder_differential_fuzz_vs_oracle(ctx):
    """1 500 mutated streams: GPU decoder vs the oracle's decoder (bytes, Err code, detail)."""
    import decode_fuzz
    for seed in (1, 2, 3):
        assert decode_fuzz.fuzz(ctx.L, ctx.h, seed=seed, cases=500) == 500


def test_compress_differential_fuzz_vs_oracle(ctx):
    """Random inputs around the block-capacity boundaries (level*100000-19), mixtures of runs / text /
    noise / periodic data, random levels: the GPU stream must equal the oracle's stream bit for bit."""
    rng = np.random.RandomState(424242)

    def piece(n):
        k = rng.randint(0, 6)
        if k == 0:
            return rng.randint(0, 256, size=n).astype(np.uint8)
        if k == 1:
            return synth.text_like(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]
        if k == 2:
            return synth.runs_mixed(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]
        if k == 3:
            return np.full(n, rng.randint(0, 256), dtype=np.uint8)
        if k == 4:
            p = rng.randint(0, 256, size=rng.randint(1, 40)).astype(np.uint8)
            return np.tile(p, n // p.size + 1)[:n]
        return synth.lcg_ascii(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]

    for case in range(160):
        level = int(rng.randint(1, 10))
        cap = level * 100000 - 19
        total = int(rng.choice([cap - 3, cap, cap + 1, cap + 255, 2 * cap + 17, 3 * cap - 1, 30011, 777, 1, 0, 1234567]))
        total = min(total, 1_400_000)
        parts, left = [], total
        while left > 0:
            n = int(min(left, rng.choice([1, 3, 4, 5, 255, 256, 1000, 50000, 99981, 300000])))
            parts.append(piece(n))
            left -= n
        d = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        got = ctx.compress(d, level)
        assert got == oracle.bz2_compress(d, level), (case, level, total)


def test_compress_periodic_and_tiled_streams_vs_oracle(ctx):
    """Whole streams that are ONE pattern tiled (periods 1 .. 65 537 bytes: constant, 'ab', 44-byte lines, RLE1's own
    4 + count period, a 30 kB text, periods around 255/256 and around the block capacity) at lengths around the block
    boundaries: every rotation of such a block ties with its neighbours for most of the block, which is the doubling rounds'
    worst case (17 full rounds) - the stream must still equal the oracle's bit for bit."""
    rng = np.random.RandomState(20260926)
    text = synth.text_like(70000, 77)
    pats = [np.array([0], np.uint8), np.frombuffer(b"ab", np.uint8), np.frombuffer(b"the quick brown fox jumps over the lazy dog\n", np.uint8),
            np.frombuffer(b"\0\0\0\0\xfb", np.uint8), rng.randint(0, 256, size=255).astype(np.uint8), rng.randint(97, 101, size=256).astype(np.uint8),
            text[:1000], text[:30011], text[:65537], np.concatenate([np.full(300, 65, np.uint8), text[:700]])]
    n = 0
    for level in (1, 2):
        cap = level * 100000 - 19
        for pat in pats:
            for total in (cap - 1, cap + 1, 2 * cap + 5, 3 * cap - 2):
                if (n + level) % 3 == 0 and pat.size > 2:       # a third of the larger cases: enough, the oracle is the slow side
                    n += 1
                    continue
                n += 1
                d = np.tile(pat, total // pat.size + 1)[:total]
                if n % 4 == 0:                                   # a foreign tail: the period breaks inside the last block
                    d = d.copy()
                    d[-37:] = rng.randint(0, 256, size=37)
                got = ctx.compress(d, level)
                assert got == oracle.bz2_compress(d, level), (level, pat.size, total)


def test_bwtc_round_trip_fuzz(ctx):
    """BWTC.compressFile -> BWTC.decompressFile on random inputs, all levels (DefSum 1-5, Fenwick 6-9);
    encoder parity itself is pinned by the 26 reference-made BWTC streams above."""
    rng = np.random.RandomState(77)
    for case in range(40):
        level = int(rng.randint(1, 10))
        n = int(rng.choice([0, 1, 2, 7, 300, 5000, 100000, level * 100000, level * 100000 + 1, 250000]))
        k = rng.randint(0, 4)
        d = (rng.randint(0, 256, size=n).astype(np.uint8) if k == 0 else synth.text_like(max(n, 1), case + 1)[:n] if k == 1
             else synth.runs_mixed(max(n, 1), case + 1)[:n] if k == 2 else np.zeros(n, np.uint8))
        z = ctx.bwtc_compress(d, level)
        assert ctx.bwtc_decompress(np.frombuffer(z, dtype=np.uint8)) == d.tobytes(), (case, level, n, k)


def test_bwtc_differential_fuzz_vs_reference(ctx):
    """308 reference-made BWTC streams (tests/golden/golden_bwtc.json, make_golden_bwtc.py): 300 small fuzz inputs over all levels
    (escape-heavy and rescale-heavy alphabets for FenwickModel, DefSumModel for 1-5), five multi-block inputs (two of them random
    bytes at levels 6 and 9: K10 emits ~1.3 % more triples than symbols there - the round-2 rows overflowed into the next block),
    and SURVEY.md 8(c)'s BWTC -9 digests of sample2/4/5."""
    import json
    import bwtc_cases
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden_bwtc.json")))["vectors"]
    n = 0
    for i in range(bwtc_cases.N_SMALL):
        d, lv = bwtc_cases.case(i)
        v = g["fuzz%d" % i]
        assert v["level"] == lv and v["in_len"] == d.size and _sha(d.tobytes()) == v["in_sha256"], i
        o = ctx.bwtc_compress(d, lv)
        assert len(o) == v["out_len"] and _sha(o) == v["out_sha256"], ("fuzz", i, lv, d.size)
        n += 1
    for cid, d, lv in bwtc_cases.big_cases():
        v = g[cid]
        o = ctx.bwtc_compress(d, lv)
        assert len(o) == v["out_len"] and _sha(o) == v["out_sha256"], cid
        assert ctx.bwtc_decompress(np.frombuffer(o, dtype=np.uint8)) == d.tobytes(), cid
        n += 1
    for cid in ("sample2", "sample4", "sample5"):
        d = cases.case_input(cid)
        assert d is not None, "reference fixtures not staged"
        v = g[cid + ":bwtc:9"]
        o = ctx.bwtc_compress(d, 9)
        assert len(o) == v["out_len"] and _sha(o) == v["out_sha256"], cid
        n += 1
    assert n == 308


def test_bwtc_model_on_gpu_equals_model_on_host(ctx):
    """K10 (FenwickModel on the GPU) against the host model of bwtc_host.hip on incompressible multi-block input at every
    level 6..9, and with K10's rows shrunk (CJS_K10_CAP) so that its overflow fall-back to the host model runs."""
    rng = np.random.RandomState(5)
    for lv in (6, 7, 8, 9):
        d = rng.randint(0, 256, size=2 * lv * 100000 + 12345).astype(np.uint8)
        a = ctx.bwtc_compress(d, lv)
        os.environ["CJS_BWTC_GPU_MODEL"] = "0"
        try:
            b = ctx.bwtc_compress(d, lv)
        finally:
            del os.environ["CJS_BWTC_GPU_MODEL"]
        os.environ["CJS_K10_CAP"] = "300000"
        try:
            c = ctx.bwtc_compress(d, lv)
        finally:
            del os.environ["CJS_K10_CAP"]
        assert a == b == c, lv
        assert ctx.bwtc_decompress(np.frombuffer(a, dtype=np.uint8)) == d.tobytes(), lv


def test_two_streams_with_host_threads_same_bytes():
    """CJS_STREAMS=2 (one host thread per stream, balanced sub-batches) must give the single-stream bytes."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "from compressjs_amd import synth\n"
        "from compressjs_amd.bzip2 import Context\n"
        "d = synth.text_like(30_000_000, 11)\n"
        "c = Context(0, 32)\n"
        "print(hashlib.sha256(c.compress(d, 9)).hexdigest(), hashlib.sha256(c.compress(d[:7_000_000], 3)).hexdigest())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for ns in ("1", "2", "3"):
        env = dict(os.environ, CJS_STREAMS=ns)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600).decode().split()[-2:])
    assert outs[0] == outs[1] == outs[2], outs


def test_periodic_blocks_closed_form_vs_oracle():
    """k1_period.hip through cjs_bwt_cyclic_batch: blocks with a linear period p <= 64 (every p, lengths that are and are not
    multiples of p, small and full alphabets, up to the -9 block capacity), near-periodic blocks and periods beyond 64 (which
    must take the general sort): transform and origPtr equal the oracle's for every block."""
    L = _lib.load()
    L.cjs_bwt_cyclic_batch.restype = C.c_int32
    L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)

    def mk(p, n, alpha):
        return np.tile(rng.integers(0, alpha, p).astype(np.uint8), n // p + 2)[:n].copy()

    def run(blocks, cap):
        nb = len(blocks)
        T = np.zeros((nb, cap), np.uint8)
        nl = np.zeros(nb, np.uint32)
        for i, d in enumerate(blocks):
            T[i, :d.size] = d
            nl[i] = d.size
        U = np.zeros((nb, cap), np.uint8)
        P = np.zeros(nb, np.uint32)
        assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
        for i, d in enumerate(blocks):
            uo, po = oracle.bwt_cyclic(d)
            assert P[i] == po and np.array_equal(U[i, :d.size], uo), (d.size, bytes(d[:70]))

    blocks = []
    for p in range(1, 65):
        n = int(rng.integers(4096, 30000))
        blocks.append(mk(p, n if p % 2 else (n // p) * p, int(rng.choice([2, 3, 4, 256]))))
    for at in (5999, 3000, 2047, 100):
        d = mk(7, 6000, 3)
        d[at] ^= 1
        blocks.append(d)
    blocks += [mk(65, 9000, 256), mk(100, 9000, 4), mk(3, 4000, 2)]
    run(blocks, 32768)
    big = [mk(1, 899981, 256), mk(2, 899981, 2), mk(5, 899980, 256), mk(44, 899981, 64), mk(64, 899981, 3), mk(63, 63 * 14000, 2)]
    d = mk(9, 899981, 4)
    d[-1] ^= 1
    big.append(d)
    # periods beyond 64 (k1p_find / k1p_reduce / k1p_expand_*): the block is sorted through its first 3 p + n mod p bytes
    text = synth.text_like(200_000, 5)
    for p, n in ((65, 899981), (8700, 899981), (8700, 8700 * 100), (70_001, 899981), (200_000, 899981), (224_995, 899981), (1000, 16384)):
        w = text[:p] if p > 300 else rng.integers(0, 3, p).astype(np.uint8)
        big.append(np.tile(w, n // p + 2)[:n].copy())
    d = np.tile(text[:50_000], 19)[:899981].copy()
    d[-1] ^= 1                                               # one foreign byte: the general sort
    big.append(d)
    run(big, 899981)
    # beyond the bzip2 block sizes (BWT.bwtransform2 takes any length): a reduced block of more than 256 tiles
    run([np.tile(synth.text_like(300_000, 6), 7)[:2_000_000].copy(), np.tile(np.frombuffer(b"abcde", np.uint8), 300_000)[:1_499_999].copy()], 2_000_000)


def _oracle_stream_digest(path):
    d = np.load(path)
    o = oracle.bz2_compress(d, 9)
    return _sha(o), len(o)


def test_data_shapes_level9_whole_streams_vs_oracle(ctx, tmp_path):
    """The 13 rows of tests/gpu_perf_probe.py (two-symbol random, text, random ASCII, runs, periodic 'ab' / 44-byte lines, zeros, random
    bytes, a 200 kB text tiled, test/sample5/4/3/2.ref tiled) at 10^7 bytes each and LEVEL 9 - the block capacity the periodic-block
    routes, the doubling rounds and the task levels are tuned for - as whole streams against the oracle (this parity lived in
    a builder-run log only); the oracle runs on the host's cores while the GPU compresses."""
    import sys
    from concurrent.futures import ProcessPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gpu_perf_probe
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        S = gpu_perf_probe.shapes(10_000_000)
    finally:
        os.chdir(cwd)
    assert len(S) >= 9
    paths = []
    for i, (_name, d) in enumerate(S):
        p = str(tmp_path / ("shape_%d.npy" % i))
        np.save(p, d)
        paths.append(p)
    with ProcessPoolExecutor(max_workers=min(len(S), os.cpu_count() or 4)) as pool:
        futs = [pool.submit(_oracle_stream_digest, p) for p in paths]
        got = [ctx.compress(d, 9) for _name, d in S]
        for (name, d), g, f in zip(S, got, futs):
            od, ol = f.result()
            assert len(g) == ol and _sha(g) == od, name


def test_adversarial_period_words_vs_oracle():
    """k1_period.hip against period words chosen to break it (tests/periodwords.py: Fibonacci and Thue-Morse prefixes, (w)^m with one
    defect, a^k b, long borders u v u, a short period with one flipped byte, tripled runs, binary noise; p in 2..64, 65..200, just below
    n / 4 and in between; n = 0, 1, p - 1 mod p) at the -9 block capacity, through cjs_bwt_cyclic_batch: transform and origPtr equal the
    oracle's, and BOTH routes - the closed form and the three-period reduction - must have been taken."""
    import subprocess
    import sys
    code = (
        "import sys, os, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import numpy as np, oracle, periodwords\n"
        "from compressjs_amd import _lib\n"
        "L = _lib.load()\n"
        "L.cjs_bwt_cyclic_batch.restype = C.c_int32\n"
        "L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]\n"
        "rng = np.random.default_rng(20260927)\n"
        "B = periodwords.blocks(899981, rng, 40)\n"
        "cap = 899981; nb = len(B)\n"
        "T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)\n"
        "for i, (_, d) in enumerate(B): T[i, :d.size] = d; nl[i] = d.size\n"
        "U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)\n"
        "assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0\n"
        "routes = L.cjs_dbg_k1_periodic_blocks()\n"
        "for i, (name, d) in enumerate(B):\n"
        "    uo, po = oracle.bwt_cyclic(d)\n"
        "    assert P[i] == po and np.array_equal(U[i, :d.size], uo), name\n"
        "print('ok', routes & 0xFFFF, routes >> 16)\n"
    ) % (ROOT, ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, CJS_K1_TRACE="1"), timeout=1500,
                                  stderr=subprocess.DEVNULL).decode().split()
    assert out[0] == "ok" and int(out[1]) >= 5 and int(out[2]) >= 10, out      # closed form / three-period reduction both taken


@pytest.mark.gpu
def test_long_period_words_same_phase_groups_vs_oracle():
    """tests/period_stress.py at the -9 block capacity: 48 blocks of period 65 .. n / 4 (text, binary noise, Q^m with defects, planted
    phrases, 0xFF-heavy words, tests/periodwords.py's families; n = 0, 1, p - 1 or anything mod p) through cjs_bwt_cyclic_batch - the
    three-period reduction and, behind it, k1d_round's same-phase shortcut - transform and origPtr equal the oracle's."""
    import subprocess
    import sys
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tests", "period_stress.py"), "41", "48"],
                                  env=dict(os.environ, CJS_STRESS_GPU="1", CJS_K1_TRACE="1"), timeout=1500, stderr=subprocess.DEVNULL).decode().split()
    assert out[0] == "ok" and out[1] == "48" and int(out[-1]) >= 4, out     # the last batch of eight went through the reduction


@pytest.mark.gpu
@pytest.mark.parametrize("env_add", [{}, {"CJS_K1_CARRY": "0"}, {"CJS_TEXT_BYTES": "0"}, {"CJS_DEEP_BIG_DIV": "1073741824"}],
                         ids=["default", "carry_off", "text_stages_off", "doubling_path_from_16"])
def test_attack_words_on_the_bucket_sort_vs_oracle(env_add):
    """tests/attackwords.py at the -9 block capacity (n = 899 981; VERDICT r5 item 6: the families the judge attacked the 16-byte-key
    bucket sort, its all-ones sentinel cells and the carried BWT byte with - words of 15 / 16 / 17 and 40..90 bytes over {0xFE,0xFF},
    {a..d} and the full alphabet, 0xFF- / 0x00-heavy blocks with 5..400 defects, one phrase of 16..64 bytes in 300..1500 places,
    near-periodic words with 1..5 flipped bits, a mixed block): 28 blocks per knob setting through cjs_bwt_cyclic_batch, transform and
    origPtr equal the oracle's."""
    import subprocess
    import sys
    code = (
        "import sys, os, ctypes as C; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import numpy as np, oracle, attackwords\n"
        "from compressjs_amd import _lib, synth\n"
        "L = _lib.load()\n"
        "L.cjs_bwt_cyclic_batch.restype = C.c_int32\n"
        "L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]\n"
        "rng = np.random.default_rng(20260930)\n"
        "B = attackwords.blocks(899981, rng, synth.text_like)\n"
        "cap = 899981; nb = len(B)\n"
        "T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)\n"
        "for i, (_, d) in enumerate(B): T[i, :d.size] = d; nl[i] = d.size\n"
        "U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)\n"
        "assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0\n"
        "for i, (name, d) in enumerate(B):\n"
        "    uo, po = oracle.bwt_cyclic(d)\n"
        "    assert P[i] == po and np.array_equal(U[i, :d.size], uo), name\n"
        "print('ok', nb)\n"
    ) % (ROOT, ROOT)
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env_add), timeout=1500,
                                  stderr=subprocess.DEVNULL).decode().split()
    assert out[0] == "ok" and int(out[1]) == 28, out


def test_deep_refinement_variants_same_bytes(ctx):
    """The text stages in front of the doubling rounds are a faster route to the same order.  K1's knobs (k1_bwt.hip,
    k1_knobs) must all give the same bytes on phrase-reuse text + runs + periodic + tiled input, and those bytes must be the
    oracle's on the leading blocks: no text stages at all (doubling from 8 bytes), the default, one refinement round only and no
    lane kernels (most ties left to the doubling rounds) on one stream, no in-bucket iteration with the predictor forcing the text
    stages on and unequal shares of a batch, the predictor forcing them off with no read-back at all, no closed form for the
    periodic blocks (k1_period.hip off: they go through the doubling rounds)."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from compressjs_amd import synth\n"
        "from compressjs_amd.bzip2 import Context\n"
        "d = np.concatenate([synth.enwik_like(16_000_000, 2025), synth.runs_mixed(1_000_000, 2), synth.periodic(900_000, b'abcab'),\n"
        "                    np.tile(synth.text_like(70_000, 3), 20)])\n"
        "c = Context(0, 32)\n"
        "print(hashlib.sha256(c.compress(d, 9)).hexdigest(), hashlib.sha256(c.compress(d[:9_000_000], 4)).hexdigest())\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for env_add in ({"CJS_TEXT_BYTES": "0"}, {}, {"CJS_TEXT_BYTES": "44", "CJS_DEEP_LANE_CAP": "0", "CJS_STREAMS": "1"},
                    {"CJS_BSORT_ITERS": "0", "CJS_DEEP_BIG_DIV": "1", "CJS_SHARES": "300:700"},
                    {"CJS_DEEP_BIG_DIV": "1000000000", "CJS_K1_SYNC": "0"}, {"CJS_K1_PERIOD": "0"}, {"CJS_K1_CARRY": "0", "CJS_NO_SIDE_CRC": "1"}):
        env = dict(os.environ, **env_add)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600).decode().split()[-2:])
    assert all(o == outs[0] for o in outs), outs
    d = synth.enwik_like(16_000_000, 2025)
    a = ctx.compress(d[:2_700_000], 9)
    assert a == oracle.bz2_compress(d[:2_700_000], 9)


def test_segmented_host_path_same_bytes(ctx):
    """cjs_bz2_compress on an input of more than 1.5 batches: segment-wise upload / plan / encode / download
    (compress_segmented) must give the bytes of the one-piece device-resident path."""
    import torch
    d = np.concatenate([synth.text_like(150_000_000, 41), synth.runs_mixed(30_000_000, 9), synth.lcg_ascii(80_000_000, 5)])
    a = ctx.compress(d, 9)
    d_in = torch.from_numpy(d).cuda()
    d_out = torch.zeros((int(ctx.L.cjs_bz2_compress_bound(d.size)) + 3) & ~3, dtype=torch.uint8, device="cuda")
    n = ctx.compress_device(d_in, d_out, 9)
    assert n == len(a) and _sha(d_out[:n].cpu().numpy().tobytes()) == _sha(a)
    assert a[:200_000] == oracle.bz2_compress(d[:3_000_000], 9)[:200_000]


def test_overlapped_host_path_same_bytes(ctx):
    """cjs_bz2_compress with CJS_SLICE_BLOCKS (compress_overlapped: upload, per-slice parallel plan, one sub-batch per slice on alternating
    streams, download behind the cursor - an option, off by default): the bytes of the one-piece path, on text, on a stream whose runs
    straddle the slice cuts (a slice that refuses its plan sends the call down the one-piece path) and at two slice sizes."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from compressjs_amd import synth\n"
        "from compressjs_amd.bzip2 import Context\n"
        "a = np.concatenate([synth.enwik_like(21_000_000, 31), synth.runs_mixed(2_000_000, 4), synth.lcg_ascii(9_000_000, 5)])\n"
        "b = np.concatenate([synth.text_like(5_390_000, 2), np.full(40_000, 7, np.uint8), synth.text_like(9_000_000, 3), np.zeros(3_000_000, np.uint8), synth.text_like(4_000_000, 8)])\n"
        "c = Context(0, 32)\n"
        "print(hashlib.sha256(c.compress(a, 9)).hexdigest(), hashlib.sha256(c.compress(b, 9)).hexdigest(), hashlib.sha256(c.compress(a[:12_000_000], 3)).hexdigest())\n"
    ) % ROOT
    outs = []
    for env_add in ({"CJS_SLICE_BLOCKS": "0"}, {"CJS_SLICE_BLOCKS": "6"}, {"CJS_SLICE_BLOCKS": "11", "CJS_STREAMS": "3"}, {"CJS_SLICE_BLOCKS": "6", "CJS_STREAMS": "1"}):
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env_add), timeout=900).decode().split()[-3:])
    assert all(o == outs[0] for o in outs), outs
    a = np.concatenate([synth.enwik_like(21_000_000, 31), synth.runs_mixed(2_000_000, 4), synth.lcg_ascii(9_000_000, 5)])
    assert _sha(ctx.compress(a[:12_000_000], 3)) == outs[0][2]


def test_multi_context_fan_out_same_bytes():
    """cjs_bz2_compress_multi (what the Node addon calls with several devices configured): three contexts - here on the
    same GPU - take the segments round-robin; windows, parallel planning (or the replicated plan), bit-shifted placement, seam bytes, trailer."""
    from compressjs_amd.bzip2 import compress_multi
    cs = [Context(0, 16) for _ in range(3)]
    one = Context(0, 128)
    L = _lib.load()
    L.cjs_dbg_multi_mallocs.restype = C.c_int
    L.cjs_dbg_multi_fallbacks.restype = C.c_int
    try:
        # Random ASCII goes through the parallel plan (every context plans its own segment).  Block 2 of THIS text stream ends inside a run
        # of 4 (byte 1 799 588): round 5 sent the call to the replicated plan for that; round 6 carries the moved boundary through the
        # segments' chain (the segments behind it are planned again: cjs_dbg_multi_replans) - no fall-back.  The zeros (a run that fills
        # blocks) still take the replicated plan: every context plans the whole input and encodes its share.  Either way all three encode.
        L.cjs_dbg_multi_replans.restype = C.c_int
        for d, lv, replicated in ((synth.enwik_like(70_000_000, 8), 9, 0), (synth.runs_mixed(40_000_000, 3), 9, None), (synth.lcg_ascii(9_000_000, 2), 1, 0),
                                  (np.zeros(50_000_000, np.uint8), 9, 1), (synth.enwik_like(40_000_000, 5), 9, 0)):
            fb, rp = L.cjs_dbg_multi_fallbacks(), L.cjs_dbg_multi_replans()
            assert _sha(compress_multi(cs, d, lv)) == _sha(one.compress(d, lv)), (d.size, lv)
            if replicated is not None:
                assert L.cjs_dbg_multi_fallbacks() - fb == replicated, (d.size, lv)
            if d.size == 70_000_000:
                assert L.cjs_dbg_multi_replans() > rp, "the boundary inside the run at byte 1 799 588 moved nothing?"
        # the per-device segment buffers are grow-only pools kept across calls: a second pass over the same inputs allocates nothing
        before = L.cjs_dbg_multi_mallocs()
        for d, lv in ((synth.enwik_like(70_000_000, 8), 9), (np.zeros(50_000_000, np.uint8), 9), (synth.lcg_ascii(9_000_000, 2), 1)):
            assert _sha(compress_multi(cs, d, lv)) == _sha(one.compress(d, lv))
        assert L.cjs_dbg_multi_mallocs() == before, "cjs_bz2_compress_multi allocated in its hot loop after warm-up"
    finally:
        for c in cs + [one]:
            c.close()


def _gloo_rank(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from compressjs_amd import synth as sy
    from compressjs_amd.bzip2 import Context as Cx
    from compressjs_amd.dist import margin_bytes, sharded_compress, sharded_compress_sliced, slice_bounds
    c = Cx(0, 32)
    d = np.concatenate([sy.text_like(9_000_000, 21), sy.runs_mixed(2_000_000, 2)])
    out = sharded_compress(c, torch.from_numpy(d).cuda(), 9)
    lo, hi = slice_bounds(d.size, rank, world)                   # the sliced driver: this rank holds [lo - margin, hi) only
    wlo = max(0, lo - margin_bytes(9))
    out2 = sharded_compress_sliced(c, torch.from_numpy(d[wlo:hi].copy()).cuda(), wlo, d.size, 9, d_all=lambda: torch.from_numpy(d).cuda())
    if rank == 0:
        assert torch.equal(out, out2)
        q.put(out.cpu().numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_real_library():
    """compressjs_amd/dist.py with the REAL library: two processes on GPU 0, gloo for the collectives (RCCL needs one
    GPU per rank): plan, block ranges, 24-byte all_gather, seam shift kernel, variable-length send/recv, assembly."""
    import torch.multiprocessing as mp
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [mctx.Process(target=_gloo_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    d = np.concatenate([synth.text_like(9_000_000, 21), synth.runs_mixed(2_000_000, 2)])
    assert got == oracle.bz2_compress(d, 9)


def test_bwtc_level9_full_size_vs_reference_digest(ctx, golden_big):
    """BASELINE.json configs[4]: BWTC -9 on the 10^8-byte E8S-A stream - BWT.bwtransform (K1 linear), MTF/RLE2 (K2) and the
    adaptive FenwickModel (K10) on the GPU, range coder on the host - against the digest of the reference's own output."""
    assert workloads.have_fixtures(), "reference fixtures not staged (run __graft_entry__.build() in the build container)"
    d = workloads.stream("e8sa", 100_000_000)
    g = golden_big["e8sa:100000000:bwtc:9"]
    a = ctx.bwtc_compress(d, 9)
    assert len(a) == g["out_len"] and _sha(a) == g["out_sha256"]


def test_lcg_generator_on_the_device(ctx):
    """cfg4's stream generated in HBM by jump-ahead (cjs_lcg_ascii_device) = synth.lcg_ascii, any slice."""
    import torch
    want = synth.lcg_ascii(3_000_000, 7)
    for first, n in ((0, 3_000_000), (123_457, 1_000_001), (2_999_999, 1)):
        t = torch.full((n + 16,), 0xEE, dtype=torch.uint8, device="cuda")
        ctx.lcg_ascii_device(t[:n], 7, first)
        got = t.cpu().numpy()
        assert np.array_equal(got[:n], want[first:first + n]) and (got[n:] == 0xEE).all()


def test_mixed_calls_on_two_threads_vs_oracle():
    """tests/gpu_stress_calls.py, bounded: 2 x 45 calls of mixed sizes (0 .. 5 MB), kinds (text, noise, runs, short periods, enwik-shaped) and
    levels through two contexts on two host threads at once - every stream up to 2 MB equals the oracle's, every stream decodes back on the
    GPU with the verdict of the oracle's decoder (libbz2 first; where it refuses - the 4th-byte-of-a-run quirk - the oracle decides)."""
    import threading
    import gpu_stress_calls
    res = []
    ts = [threading.Thread(target=gpu_stress_calls.work, args=(sd, 45, res)) for sd in (5, 6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(res) == 2 and sum(r[0] for r in res) == 0, res
