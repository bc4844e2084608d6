"""BWTC -9 (BASELINE.json configs[4]) rate, host buffers in and out (not a test): python tests/gpu_bwtc_probe.py [workload] [size]
CJS_BWTC_GPU_MODEL=0 runs the adaptive model on the host as in round 1."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import workloads
from compressjs_amd.bzip2 import Context
wl = sys.argv[1] if len(sys.argv) > 1 else 'e8sa'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
d = workloads.stream(wl, n)
ctx = Context(0, 128)
tt = []
for _ in range(3):
    a = time.perf_counter(); out = ctx.bwtc_compress(d, 9); tt.append(time.perf_counter() - a)
print('BWTC -9 %s %d B -> %d B: %.3f s = %.1f MB/s (GPU stages %.1f ms; model on %s) sha %s' % (
    wl, n, len(out), min(tt), n / min(tt) / 1e6, ctx.last_device_ms, 'host' if os.environ.get('CJS_BWTC_GPU_MODEL') == '0' else 'GPU',
    hashlib.sha256(out).hexdigest()[:16]), flush=True)
if '--roundtrip' in sys.argv:
    print('roundtrip', ctx.bwtc_decompress(out) == d.tobytes())
