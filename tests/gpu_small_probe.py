"""Latency probe of a small input (not a test): BASELINE.json configs[1], test/sample5.ref (2 130 640 bytes, three blocks) -9 through cjs_bz2_compress
(host buffer in, host buffer out).  python tests/gpu_small_probe.py [reps]   - under rocprofv3 --kernel-trace for the kernel timeline of the last call."""
import sys, os, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from compressjs_amd.bzip2 import Context
p = os.path.join(ROOT, 'oracle', '_ref', 'fixtures', 'sample5.ref')
d = np.fromfile(p, dtype=np.uint8)
ctx = Context(0, 128)
out = np.zeros(int(ctx.L.cjs_bz2_compress_bound(d.size)), np.uint8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3):
    n = int(ctx.L.cjs_bz2_compress(ctx.h, d.ctypes.data, d.size, 9, out.ctypes.data, out.size))
t = []
for _ in range(reps):
    a = time.perf_counter()
    n = int(ctx.L.cjs_bz2_compress(ctx.h, d.ctypes.data, d.size, 9, out.ctypes.data, out.size))
    t.append(time.perf_counter() - a)
print('sample5.ref -9: %d -> %d bytes, %.3f ms best, %.3f ms median, %.1f MB/s, sha %s' % (d.size, n, min(t) * 1e3, sorted(t)[len(t) // 2] * 1e3, d.size / min(t) / 1e6, hashlib.sha256(out[:n].tobytes()).hexdigest()[:12]))
