"""Host buffer in -> .bz2 in host memory (cjs_bz2_compress) against the device-resident step, for CJS_SEG_BYTES settings (not a test):
python tests/gpu_host_path_probe.py [workload]"""
import sys, os, time, hashlib, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import numpy as np, torch, workloads
    from compressjs_amd.bzip2 import Context
    w = sys.argv[1]
    host = workloads.stream(w, 100_000_000)
    ctx = Context(0, 128)
    bound = int(ctx.L.cjs_bz2_compress_bound(host.size))
    hbuf = np.zeros(bound, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda(); d_out = torch.zeros((bound + 3) & ~3, dtype=torch.uint8, device='cuda')
    for _ in range(3): n = ctx.compress_device(d_in, d_out, 9)
    dev = min(ctx.last_device_ms for _ in range(3) if ctx.compress_device(d_in, d_out, 9))
    tt = []
    for _ in range(6):
        a = time.perf_counter()
        nn = int(ctx.L.cjs_bz2_compress(ctx.h, host.ctypes.data, host.size, 9, hbuf.ctypes.data, hbuf.size))
        tt.append(time.perf_counter() - a)
    print('%-28s %s device %.2f ms; host-to-host best %.2f ms = %.0f MB/s  (ratio %.2f)  sha %s' % (
        os.environ.get('CJS_SEG_BYTES', 'one piece'), w, dev, min(tt[1:]) * 1e3, host.size / min(tt[1:]) / 1e6, dev / (min(tt[1:]) * 1e3),
        hashlib.sha256(hbuf[:nn].tobytes()).hexdigest()[:12]), flush=True)
else:
    w = sys.argv[1] if len(sys.argv) > 1 else 'enwik'
    for seg in (None, '50000000', '34000000', '25000000'):
        env = dict(os.environ)
        if seg: env['CJS_SEG_BYTES'] = seg
        subprocess.call([sys.executable, os.path.abspath(__file__), w, 'child'], env=env)
