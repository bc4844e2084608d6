"""Timing sanity over different data shapes (not a test): python tests/gpu_perf_probe.py"""
import sys, os, time, bz2
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
ctx = Context(0, 128)
def run(name, data, check=True):
    d_in = torch.from_numpy(data).cuda()
    cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
    d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
    import hashlib
    n = ctx.compress_device(d_in, d_out, 9)
    t = []; hs = set()
    for _ in range(4):
        n = ctx.compress_device(d_in, d_out, 9); t.append(ctx.last_device_ms)
        hs.add(hashlib.sha256(d_out[:n].cpu().numpy().tobytes()).hexdigest()[:12])
    if len(hs) > 1: print('   NONDETERMINISTIC OUTPUT', name, hs, flush=True)
    ok = ''
    if check:
        try:
            ok = bz2.decompress(d_out[:n].cpu().numpy().tobytes()) == data.tobytes()
        except Exception as e:
            ok = 'DECODE ERROR ' + repr(e)
    print('%-28s %10d -> %9d  %8.2f ms  %8.1f MB/s  blocks %d rounds %d sparse %d roundtrip %s' % (name, data.size, n, min(t), data.size / min(t) / 1e3, ctx.last_block_count, ctx.L.cjs_dbg_k1_rounds(), ctx.L.cjs_dbg_k1_sparse_rounds(), ok), flush=True)
N = 50_000_000
rng = np.random.RandomState(1)
_ = rng.randint(0, 256, size=N)
two = rng.randint(97, 99, size=N).astype(np.uint8)
run('two-symbol random', two)
run('text_like', synth.text_like(N, 2025))
run('lcg_ascii', synth.lcg_ascii(N, 7))
run('runs_mixed', synth.runs_mixed(N, 3))
run('periodic ab', synth.periodic(N, b'ab'))
run('periodic 44B', synth.periodic(N, b'the quick brown fox jumps over the lazy dog\n'))
run('zeros', np.zeros(N, np.uint8))
rng = np.random.RandomState(1)
run('random bytes', rng.randint(0, 256, size=N).astype(np.uint8))
base = synth.text_like(200_000, 5)
run('200k text tiled', np.tile(base, N // base.size))
run('two-symbol random', two)
for f in ('sample5.ref', 'sample4.ref', 'sample3.ref', 'sample2.ref'):
    p = os.path.join('oracle', '_ref', 'fixtures', f)
    if os.path.exists(p):
        d = np.fromfile(p, dtype=np.uint8)
        run(f + ' tiled', np.tile(d, max(1, N // d.size)))
