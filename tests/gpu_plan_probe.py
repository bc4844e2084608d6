"""Cost of the block-split pre-pass (cjs_bz2_plan) and of the 8-rank sharding steps, measured on one GPU:
python tests/gpu_plan_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
from compressjs_amd import dist as cdist
ctx = Context(0, 128)
base = synth.text_like(100_000_000, 2025)
for mult in (1, 2, 4, 8):
    d = torch.from_numpy(np.tile(base, mult)).cuda()
    ctx.plan(d, 9)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        nb = ctx.plan(d, 9)
    torch.cuda.synchronize()
    print("plan of %d bytes: %.3f ms (%d blocks)" % (d.numel(), (time.perf_counter() - t) / 5 * 1e3, nb), flush=True)
    del d
# the per-rank extras of sharded_compress on a 38 MB segment
seg = torch.randint(0, 255, (38_000_000,), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    s = cdist.shift_bits(seg, seg.numel() - 8, 3)
torch.cuda.synchronize()
print("shift_bits 38 MB: %.3f ms" % ((time.perf_counter() - t) / 5 * 1e3))
final = torch.zeros(8 * 38_000_000 + 64, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    final.zero_()
    for r in range(8):
        final[r * 38_000_000:r * 38_000_000 + s.numel()] |= s
torch.cuda.synchronize()
print("assemble 8 x 38 MB: %.3f ms" % ((time.perf_counter() - t) / 5 * 1e3))
