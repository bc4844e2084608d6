"""Host-buffer path (cjs_bz2_compress) vs device-resident (not a test): python tests/gpu_host_probe2.py [sizes...]
Prints wall time of the C-ABI call (pageable numpy memory in and out), the device time of the same stream resident
in HBM, and checks that both give the same bytes."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import workloads
from compressjs_amd.bzip2 import Context
sizes = [int(float(a)) for a in sys.argv[1:] if not a.startswith('-')] or [100_000_000, 400_000_000]
wl = 'text'
ctx = Context(0, 128)
for n in sizes:
    host = workloads.stream(wl, n)
    cap = int(ctx.L.cjs_bz2_compress_bound(n))
    out = np.zeros(cap, np.uint8)
    tt = []
    for _ in range(4):
        a = time.perf_counter()
        m = int(ctx.L.cjs_bz2_compress(ctx.h, host.ctypes.data, host.size, 9, out.ctypes.data, cap))
        tt.append(time.perf_counter() - a)
    assert m > 0, m
    sha = hashlib.sha256(out[:m].tobytes()).hexdigest()
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
    td = []
    for _ in range(3):
        k = ctx.compress_device(d_in, d_out, 9); td.append(ctx.last_device_ms)
    same = hashlib.sha256(d_out[:k].cpu().numpy().tobytes()).hexdigest() == sha and k == m
    print('%s %10d B: host path %.2f ms = %.0f MB/s (runs %s); device-resident %.2f ms = %.0f MB/s; ratio %.2f; same bytes %s' % (
        wl, n, min(tt[1:]) * 1e3, n / min(tt[1:]) / 1e6, ' '.join('%.1f' % (t * 1e3) for t in tt), min(td), n / min(td) / 1e3,
        min(td) / (min(tt[1:]) * 1e3), same), flush=True)
    del d_in, d_out
