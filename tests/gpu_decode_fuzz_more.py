"""More of the differential decoder fuzz on the GPU box (not a test): python tests/gpu_decode_fuzz_more.py [first_seed] [seeds] [cases]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import decode_fuzz
from compressjs_amd.bzip2 import Context
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 500
ctx = Context(0, 128)
t = time.time()
for seed in range(first, first + seeds):
    assert decode_fuzz.fuzz(ctx.L, ctx.h, seed=seed, cases=cases) == cases
    print("seed", seed, "ok", round(time.time() - t, 1), "s", flush=True)
