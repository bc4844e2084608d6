"""Host buffer in -> .bz2 in host memory (cjs_bz2_compress) against the device-resident step, for CJS_SLICE_BLOCKS settings (not a test):
python tests/gpu_host_path_probe.py [workload] [size]"""
import sys, os, time, hashlib, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
if len(sys.argv) > 3 and sys.argv[3] == "child":
    import numpy as np, torch, workloads
    from compressjs_amd.bzip2 import Context
    w, size = sys.argv[1], int(sys.argv[2])
    host = workloads.stream(w, size)
    ctx = Context(0, 128)
    bound = int(ctx.L.cjs_bz2_compress_bound(host.size))
    hbuf = np.zeros(bound, dtype=np.uint8)
    d_in = torch.from_numpy(host).cuda(); d_out = torch.zeros((bound + 3) & ~3, dtype=torch.uint8, device='cuda')
    for _ in range(3): n = ctx.compress_device(d_in, d_out, 9)
    dev = min(ctx.last_device_ms for _ in range(4) if ctx.compress_device(d_in, d_out, 9))
    want = hashlib.sha256(d_out[:n].cpu().numpy().tobytes()).hexdigest()[:12]
    tt = []
    for _ in range(8):
        a = time.perf_counter()
        nn = int(ctx.L.cjs_bz2_compress(ctx.h, host.ctypes.data, host.size, 9, hbuf.ctypes.data, hbuf.size))
        tt.append(time.perf_counter() - a)
    got = hashlib.sha256(hbuf[:max(nn, 0)].tobytes()).hexdigest()[:12]
    print('slice_blocks=%-8s %s %d B: device %.2f ms; host-to-host best %.2f ms (mean of last 6 %.2f) = %.0f MB/s  (ratio %.2f)  rc %d sha %s %s' % (
        os.environ.get('CJS_SLICE_BLOCKS', 'default'), w, size, dev, min(tt[1:]) * 1e3, sum(tt[2:]) / 6 * 1e3, host.size / min(tt[1:]) / 1e6, dev / (min(tt[1:]) * 1e3),
        nn, got, 'EQUAL' if got == want else 'DIFFERENT from the device path ' + want), flush=True)
else:
    w = sys.argv[1] if len(sys.argv) > 1 else 'enwik'
    size = sys.argv[2] if len(sys.argv) > 2 else '100000000'
    for sb in os.environ.get('SLICES', '0 28 20 14 40').split():
        env = dict(os.environ, CJS_SLICE_BLOCKS=sb)
        subprocess.call([sys.executable, os.path.abspath(__file__), w, size, 'child'], env=env)
