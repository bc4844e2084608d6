"""The allocator entry (cjs_huff_lengths_batch: one lane builds the parent pointers, the wave turns them into depths, both
branches of the length limiter) against the oracle's restatement of lib/HuffmanAllocator.js on random sorted weight vectors:
flat, geometric (deep trees: the relocation branch), Fibonacci-like, many equal weights, zeros; lengths 3 .. 258 and a few
long ones; maximum lengths 15 .. 32.  Kernel logic through the CPU debug build."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _vectors(rng, count):
    out = []
    for i in range(count):
        n = int(rng.choice([3, 4, 5, 7, 16, 33, 64, 65, 100, 128, 200, 257, 258, 300, 700]))
        kind = i % 6
        if kind == 0:
            f = rng.randint(0, 1000, size=n)
        elif kind == 1:
            f = np.minimum(rng.rand(n) * 1.35 ** np.minimum(np.arange(n), 90), 2.0 ** 40).astype(np.int64)
        elif kind == 2:
            f = np.cumsum(rng.randint(0, 3, size=n))
        elif kind == 3:
            a, b, f = 1, 1, []
            for _ in range(n):
                f.append(a % (1 << 45)); a, b = b, a + b
            f = np.array(f)
        elif kind == 4:
            f = rng.choice([0, 1, 2, 50000], size=n, p=[0.3, 0.4, 0.2, 0.1])
        else:
            f = (2.0 ** (rng.rand(n) * 30)).astype(np.int64)
        out.append(np.sort(np.asarray(f, dtype=np.int64)))
    return out


def test_allocator_wave_path_vs_oracle_random_vectors():
    import oracle
    import stagelib
    from compressjs_amd import _lib
    stagelib.build_emu()
    L = _lib.load(stagelib.EMU_SO)
    rng = np.random.RandomState(31337)
    bad = 0
    for ml in (15, 17, 20, 32):       # (far below log2(len) + a few the reference itself indexes outside its array: not a domain to compare in)
        vs = [v for v in _vectors(rng, 240) if v.size <= (1 << ml)]
        off = np.zeros(len(vs) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([v.size for v in vs])
        flat = np.concatenate(vs).astype(np.int64)
        assert L.cjs_huff_lengths_batch(flat.ctypes.data, off.ctypes.data, len(vs), ml) == 0
        for k, v in enumerate(vs):
            want = oracle.huff_lengths(v, ml)
            got = flat[off[k]:off[k + 1]].tolist()
            assert got == want, (ml, k, v.size)
            bad += max(want) == ml
    assert bad > 50          # the length limit was reached (relocation branch) in a good share of the cases
