"""Inputs of the BWTC differential fuzz (tests/golden/golden_bwtc.json holds the REFERENCE's output digests for them, made by
make_golden_bwtc.py under node in the build container).  Case i is a pure function of i, so the tests rebuild the inputs
instead of storing them.  The shapes aim at the branches of lib/FenwickModel.js:47-87,137-172 that the text fixtures
rarely take: escapes of symbols that were scaled down to zero (large alphabets, long gaps between occurrences), the last
escape (root low16 == 1: tiny alphabets that fill up), rescales (every ~127 symbols at increment 0x100), and DefSumModel
(levels 1-5); a few span several blocks (level * 100000 bytes, lib/BWTC.js:27)."""
import numpy as np

N_SMALL = 300


def _gap_heavy(rng, n):
    """mostly two symbols, every ~1200 bytes a sweep over rarely used ones: every sweep symbol has been scaled to 0 -> escapes again"""
    out = rng.randint(97, 99, size=n).astype(np.uint8)
    rare = np.arange(160, 250, dtype=np.uint8)
    p = 900
    while p + rare.size < n:
        k = int(rng.randint(8, rare.size))
        out[p:p + k] = rng.permutation(rare)[:k]
        p += int(rng.randint(1000, 1600))
    return out


def case(i: int):
    """-> (data uint8 array, level)"""
    rng = np.random.RandomState(1000 + i)
    level = 1 + (i * 7 + i // 9) % 9
    style = i % 10
    n = int(rng.choice([1, 2, 3, 17, 200, 1000, 2500, 6000, 12000, 30000]))
    if style == 0:
        d = rng.randint(0, 256, size=n).astype(np.uint8)                       # all 256 symbols, uniform
    elif style == 1:
        d = rng.randint(0, int(rng.choice([2, 3, 5, 17])), size=n).astype(np.uint8)   # tiny alphabets: the escape leaf runs dry
    elif style == 2:
        d = _gap_heavy(rng, max(n, 4000))
    elif style == 3:
        d = np.repeat(rng.randint(0, 256, size=max(1, n // 50)).astype(np.uint8), rng.randint(1, 100, size=max(1, n // 50)))[:max(n, 1)]   # runs
    elif style == 4:
        words = [bytes(rng.randint(97, 123, size=int(rng.randint(2, 9))).astype(np.uint8)) for _ in range(40)]
        d = np.frombuffer(b" ".join(words[int(j)] for j in rng.randint(0, 40, size=max(1, n // 5))), dtype=np.uint8)[:max(n, 1)].copy()
    elif style == 5:
        d = (np.cumsum(rng.randint(-2, 3, size=n)) & 255).astype(np.uint8)     # slowly drifting values: MTF indices stay small
    elif style == 6:
        d = np.tile(rng.randint(0, 256, size=int(rng.randint(1, 60))).astype(np.uint8), max(1, n // 20))[:max(n, 1)]   # periodic
    elif style == 7:
        d = rng.choice(np.array([0, 1, 2, 254, 255], dtype=np.uint8), size=n, p=[0.9, 0.04, 0.03, 0.02, 0.01])      # skewed
    elif style == 8:
        d = np.concatenate([rng.randint(0, 256, size=n // 2 + 1), np.zeros(n // 2 + 1, dtype=np.int64), rng.randint(0, 4, size=n // 2 + 1)]).astype(np.uint8)
    else:
        d = np.sort(rng.randint(0, 256, size=n)).astype(np.uint8)              # every symbol once in a row: first occurrences only
    return np.ascontiguousarray(d), level


def big_cases():
    """multi-block inputs: (id, data, level)"""
    rng = np.random.RandomState(77)
    out = []
    out.append(("rand250k_l1", rng.randint(0, 256, size=250_000).astype(np.uint8), 1))
    out.append(("rand250k_l2", rng.randint(0, 256, size=250_000).astype(np.uint8), 2))
    out.append(("gap1300k_l6", _gap_heavy(rng, 1_300_000), 6))                 # two full blocks + a tail at level 6
    out.append(("rand1250k_l6", rng.randint(0, 256, size=1_250_000).astype(np.uint8), 6))   # K10: ~1.3 % more triples than symbols
    out.append(("rand1850k_l9", rng.randint(0, 256, size=1_850_000).astype(np.uint8), 9))
    return out
