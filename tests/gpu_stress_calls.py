"""Many calls of mixed sizes and levels through one context per thread (not a test: [CJS_STRESS_SEEDS=1,2] python tests/gpu_stress_calls.py [n]):
every stream up to 2 MB of input is compared with the oracle's; every stream is decoded back by the GPU decoder, and its verdict must be the ORACLE decoder's (the
restatement of the reference's Bunzip) - libbz2 is consulted first because it is faster, and where it refuses the oracle decides (a block that fills on the 4th byte of
a run gets no count byte from the reference - lib/Bzip2.js:640-644, mirrored on purpose: the reference's own decoder takes such a stream, libbz2 reports a CRC error)."""
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
        if not valid and size:
            valid = oracle.bz2_decompress(np.frombuffer(c, dtype=np.uint8))[2] == d.tobytes()     # the arbiter: the reference's decoder, restated
        try:
            back = ctx.decompress(np.frombuffer(c, dtype=np.uint8)) if size else b""
            ok = bytes(back) == d.tobytes()
        except Exception:
            ok = False
        if size and ok != valid: bad += 1; print("DECODER", seed, k, size, level, kind, "libbz2 valid", valid, "gpu round trip", ok, flush=True)
    out.append((bad, n))

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    res = []
    seeds = tuple(int(x) for x in os.environ.get('CJS_STRESS_SEEDS', '1,2').split(','))        # one thread (and context) per seed
    ts = [threading.Thread(target=work, args=(s, n, res)) for s in (seeds if len(sys.argv) < 3 else seeds[:1])]
    for t in ts: t.start()
    for t in ts: t.join()
    print("threads finished", len(res), "calls", sum(r[1] for r in res), "bad", sum(r[0] for r in res))
