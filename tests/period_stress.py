"""Stress of the three-period reduction and of the same-phase shortcut in k1d_round (not collected by pytest): blocks T[i] = P[i mod p] with 64 < p <= n / 4
whose period word is text, binary noise, a word with inner repeats or one of tests/periodwords.py's families, at n = 0, 1, p - 1 or a random residue (mod p),
against the oracle's cyclic BWT.  python tests/period_stress.py [seed] [blocks] runs the CPU logic-debug build; CJS_STRESS_GPU=1 the HIP library at the -9 capacity."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle, periodwords
from compressjs_amd import synth
GPU = os.environ.get("CJS_STRESS_GPU") == "1"
if GPU:
    from compressjs_amd import _lib
    L = _lib.load()
else:
    import stagelib
    L = C.CDLL(stagelib.EMU_SO)
L.cjs_bwt_cyclic_batch.restype = C.c_int32
L.cjs_bwt_cyclic_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
cap = 899981 if GPU else 120000
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 24


def word(p):
    k = int(rng.integers(0, 6))
    if k == 0:
        return "text", synth.text_like(p, int(rng.integers(1, 1 << 20)))
    if k == 1:
        return "binary", rng.integers(97, 99, p).astype(np.uint8)
    if k == 2:                                                            # inner repeats: Q^m with a few defects, so groups hold several phases
        q = int(rng.integers(3, max(4, p // 3)))
        d = np.tile(rng.integers(97, 101, q).astype(np.uint8), p // q + 1)[:p].copy()
        d[rng.integers(0, p, int(rng.integers(1, 4)))] ^= 1
        return "inner q=%d" % q, d
    if k == 3:                                                            # one phrase planted many times in the period
        d = rng.integers(97, 123, p).astype(np.uint8)
        ph = rng.integers(97, 123, int(rng.integers(8, 80))).astype(np.uint8)
        if p > ph.size + 1:
            for s in rng.integers(0, p - ph.size, int(rng.integers(2, 40))):
                d[s:s + ph.size] = ph
        return "phrase", d
    if k == 4:
        return "ff", np.where(rng.random(p) < 0.9, 255, rng.integers(0, 256, p)).astype(np.uint8)
    ws = periodwords.period_words(p, rng)
    return ws[int(rng.integers(0, len(ws)))]


B = []
while len(B) < nblk:
    n0 = int(rng.integers(cap // 3, cap + 1))
    p = int(rng.integers(65, n0 // 4)) if rng.random() < 0.7 else int(rng.integers(4200, max(4201, n0 // 4)))
    name, w = word(p)
    res = (0, 1, p - 1, int(rng.integers(0, p)))[int(rng.integers(0, 4))]
    n = (n0 // p) * p + res
    if n > cap:
        n -= p
    if n < 4 * p:
        continue
    B.append(("%s p=%d n=%d r0=%d" % (name, p, n, n % p), np.tile(w, n // p + 2)[:n].copy()))
for i in range(0, len(B), 8):
    blocks = B[i:i + 8]; nb = len(blocks)
    T = np.zeros((nb, cap), np.uint8); nl = np.zeros(nb, np.uint32)
    for j, (_, d) in enumerate(blocks): T[j, :d.size] = d; nl[j] = d.size
    U = np.zeros((nb, cap), np.uint8); P = np.zeros(nb, np.uint32)
    assert L.cjs_bwt_cyclic_batch(T.ctypes.data, nl.ctypes.data, nb, cap, U.ctypes.data, P.ctypes.data) == 0
    for j, (name, d) in enumerate(blocks):
        uo, po = oracle.bwt_cyclic(d)
        assert P[j] == po and (U[j, :d.size] == uo).all(), name
routes = L.cjs_dbg_k1_periodic_blocks()
print('ok', nblk, 'closed form', routes & 0xFFFF, 'reduced', routes >> 16)
