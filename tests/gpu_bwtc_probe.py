"""BWTC -9 timing (BASELINE.json configs[4]): python tests/gpu_bwtc_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
ctx = Context(0, 128)
for n in (20_000_000, 100_000_000):
    d = synth.text_like(n, 2025)
    ctx.bwtc_compress(d[:2_000_000], 9)
    t = time.perf_counter(); o = ctx.bwtc_compress(d, 9); dt = time.perf_counter() - t
    print('BWTC -9 text %d -> %d bytes: wall %.3f s = %.1f MB/s (GPU stages + serial host range coder; last_ms %.1f)' % (n, len(o), dt, n / dt / 1e6, ctx.last_device_ms), flush=True)
