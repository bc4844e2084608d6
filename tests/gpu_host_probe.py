"""Host-buffer (PCIe-inclusive) rate of cjs_bz2_compress: python tests/gpu_host_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
ctx = Context(0, 128)
d = synth.text_like(100_000_000, 2025)
ctx.compress(d[:5_000_000], 9)
import numpy as _np
cap = int(ctx.L.cjs_bz2_compress_bound(d.size)); obuf = _np.zeros(cap, dtype=_np.uint8)
for _ in range(3):
    t = time.perf_counter(); o = ctx.compress(d, 9); dt = time.perf_counter() - t
    t = time.perf_counter(); n = ctx.L.cjs_bz2_compress(ctx.h, d.ctypes.data, d.size, 9, obuf.ctypes.data, cap); dc = time.perf_counter() - t
    print('host-buffer compress 1e8 B: C ABI call %.1f ms (device part %.1f ms) -> %.0f MB/s PCIe-inclusive; Python wrapper incl. bytes copy %.1f ms' % (dc * 1e3, ctx.last_device_ms, 1e8 / dc / 1e6, dt * 1e3), flush=True)

# K6: BWT.unbwtransform on a 9e5-byte block (host buffers; includes H2D/D2H and allocation)
import numpy as np
from compressjs_amd.bzip2 import BWT
blk = d[:900_000]
U = np.zeros(blk.size, np.uint8)
p = BWT.bwtransform(blk, U, None, blk.size)
back = np.zeros(blk.size, np.uint8)
for _ in range(3):
    t = time.perf_counter(); BWT.unbwtransform(U, back, None, blk.size, p); dt = time.perf_counter() - t
    print('unbwtransform 9e5 B: wall %.2f ms, ok=%s' % (dt * 1e3, bool((back == blk).all())), flush=True)
