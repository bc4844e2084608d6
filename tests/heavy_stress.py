"""Stress of the heavy-key sub-buckets of k1_front.hip on the CPU logic-debug build (not collected by pytest): HTML-like blocks - a few 8-byte prefixes in 60 % of the
lines, their continuations from uniform to 99 % one value, 0xFF sprinkled in - against the oracle's cyclic BWT.  CJS_DEEP_BIG_DIV=1073741824 python tests/heavy_stress.py [seed] [blocks]
forces the path without text stages (sub-buckets, buckets of one 16-byte key as groups, doubling rounds from h = 16)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle, stagelib
from compressjs_amd import synth
L = C.CDLL(stagelib.EMU_SO)
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
def html_like(n):
    # heavy 8-byte prefixes with continuations of varied skew: some dominated by one continuation (16-byte heavy), some spread
    keys = [bytes(rng.integers(97, 123, 8).astype(np.uint8)) for _ in range(int(rng.integers(2, 12)))]
    conts = {k: [bytes(rng.integers(32, 127, int(rng.integers(8, 30))).astype(np.uint8)) for _ in range(int(rng.integers(1, 40)))] for k in keys}
    skew = {k: float(rng.choice([0.0, 0.5, 0.9, 0.99])) for k in keys}
    parts, tot = [], 0
    while tot < n:
        if rng.random() < 0.6:
            k = keys[int(rng.integers(0, len(keys)))]
            c = conts[k][0] if rng.random() < skew[k] else conts[k][int(rng.integers(0, len(conts[k])))]
            p = k + c
        else:
            p = bytes(rng.integers(32, 127, int(rng.integers(5, 60))).astype(np.uint8))
        parts.append(p); tot += len(p)
    d = np.frombuffer(b''.join(parts), np.uint8)[:n].copy()
    if rng.random() < 0.3:
        d[rng.integers(0, n, n // 50)] = 255                      # sprinkle 0xFF (the all-ones sentinels)
    return d
cap = 160000
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = [html_like(int(rng.integers(30000, cap + 1))) for _ in range(nblk)]
for i in range(0, len(B), 8):
    blocks = B[i:i + 8]; nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for j, d in enumerate(blocks): T[j, :d.size] = d; nl[j] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    for j, d in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[j] == po and (U[j, :d.size] == uo).all(), (i + j, d.size)
print('ok', nblk)
