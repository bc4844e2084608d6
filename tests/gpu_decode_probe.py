"""Decode-rate probe on the GPU box: python tests/gpu_decode_probe.py [bytes]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0, 128)
d = synth.text_like(n, 2025)
z = np.frombuffer(ctx.compress(d, 9), dtype=np.uint8)
zi = torch.from_numpy(z.copy()).cuda()
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter(); m = ctx.decompress_device(zi, out); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("decompress_device: %d -> %d bytes, wall %.1f ms (events %.1f ms) -> %.0f MB/s of output" % (z.size, m, dt * 1e3, ctx.last_decode_ms, m / dt / 1e6), flush=True)
assert m == n and bool((out[:n].cpu().numpy() == d).all())
print("round trip ok")
if n >= 100_000_000:
    print("blocks in flight: min(%d candidates, 4096); 4 waves per block" % ((n + 899980) // 899981))
