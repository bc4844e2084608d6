"""A/B of variant libraries on a few data shapes (not a test): COMPRESSJS_AMD_LIB=... python tests/gpu_r6_shapes_ab.py [names...]"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import gpu_perf_probe as P
from compressjs_amd.bzip2 import Context
want = sys.argv[1:]
ctx = Context(0, 128)
for name, data in P.shapes():
    if want and not any(w in name for w in want):
        continue
    d_in = torch.from_numpy(data).cuda()
    cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
    d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
    t = []
    for _ in range(5):
        n = ctx.compress_device(d_in, d_out, 9); t.append(ctx.last_device_ms)
    print('%-22s %8.2f ms %9.1f MB/s sha %s' % (name, min(t[1:]), data.size / min(t[1:]) / 1e3, hashlib.sha256(d_out[:n].cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
