"""Many calls of mixed sizes and levels through one context and through two contexts on two Python threads (not a test: python tests/gpu_stress_calls.py [n]):
every stream up to 2 MB of input is compared with the oracle's; every stream libbz2 accepts is decoded back by the GPU decoder and compared with the input (a block
that fills on the 4th byte of a run gets no count byte from the reference - lib/Bzip2.js:640-644, mirrored on purpose - and neither libbz2 nor the reference's own
decoder take such a stream: the GPU decoder must refuse it too)."""
import sys, os, threading, bz2
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import oracle
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context

def work(seed, n, out):
    rng = np.random.RandomState(seed)
    ctx = Context(0, 128)
    bad = 0
    for k in range(n):
        size = int(rng.choice([0, 1, 777, 30011, 99981, 100001, 250000, 899981, 900123, 2_000_000, 5_000_000]))
        kind = k % 5
        if kind == 0: d = synth.text_like(max(size, 1), seed * 1000 + k)[:size]
        elif kind == 1: d = rng.randint(0, 256, size=size).astype(np.uint8)
        elif kind == 2: d = synth.runs_mixed(max(size, 1), k)[:size]
        elif kind == 3: d = np.tile(rng.randint(0, 4, size=int(rng.randint(1, 3000))).astype(np.uint8), size // 1 + 1)[:size]
        else: d = synth.enwik_like(max(size, 1), k)[:size]
        d = np.ascontiguousarray(d)
        level = int(rng.randint(1, 10))
        c = ctx.compress(d, level)
        if size <= 2_000_000 and c != oracle.bz2_compress(d, level): bad += 1; print("ORACLE", seed, k, size, level, flush=True)
        try:
            valid = bz2.decompress(c) == d.tobytes()
        except OSError:
            valid = False
        try:
            back = ctx.decompress(np.frombuffer(c, dtype=np.uint8)) if size else b""
            ok = bytes(back) == d.tobytes()
        except Exception:
            ok = False
        if size and ok != valid: bad += 1; print("DECODER", seed, k, size, level, kind, "libbz2 valid", valid, "gpu round trip", ok, flush=True)
    out.append((bad, n))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
res = []
ts = [threading.Thread(target=work, args=(s, n, res)) for s in ((1, 2) if len(sys.argv) < 3 else (1,))]
for t in ts: t.start()
for t in ts: t.join()
print("threads finished", len(res), "calls", sum(r[1] for r in res), "bad", sum(r[0] for r in res))
