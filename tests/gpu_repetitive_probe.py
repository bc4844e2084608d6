"""Kernel-level look at the worst case of prefix doubling (periodic input): python tests/gpu_repetitive_probe.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
ctx = Context(0, 128)
which = sys.argv[1] if len(sys.argv) > 1 else "periodic44"
N = 50_000_000
d = synth.periodic(N, b'the quick brown fox jumps over the lazy dog\n') if which == "periodic44" else synth.periodic(N, b'ab')
d_in = torch.from_numpy(d).cuda()
d_out = torch.zeros(int(ctx.L.cjs_bz2_compress_bound(N)), dtype=torch.uint8, device='cuda')
for _ in range(2):
    n = ctx.compress_device(d_in, d_out, 9)
    print(which, n, ctx.last_device_ms, "ms", ctx.L.cjs_dbg_k1_rounds(), "rounds")
