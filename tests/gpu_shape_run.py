"""One shape of tests/gpu_perf_probe.py, compressed a few times (for rocprofv3 runs): python tests/gpu_shape_run.py <substring of the shape's name> [reps]"""
import sys, os, hashlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
os.chdir(os.path.dirname(HERE))
import numpy as np, torch
import gpu_perf_probe
from compressjs_amd.bzip2 import Context

want = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = Context(0, 128)
for name, data in gpu_perf_probe.shapes():
    if want not in name:
        continue
    d_in = torch.from_numpy(data).cuda()
    cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
    d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
    t = []
    for _ in range(reps):
        n = ctx.compress_device(d_in, d_out, 9); t.append(ctx.last_device_ms)
    print('%-22s %9d -> %9d  %8.2f ms  %s' % (name, data.size, n, min(t), hashlib.sha256(d_out[:n].cpu().numpy().tobytes()).hexdigest()[:12]), flush=True)
