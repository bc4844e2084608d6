"""A/B probe for K1-deep (not a test): python tests/gpu_deep_probe.py gen | run <name>...
`gen` caches the synthetic streams under /tmp; `run` compresses each 4x and prints the best device time, the
K1 round counts and a digest of the stream, which must not depend on CJS_DEEP_ITERS / CJS_DEEP_TILE."""
import sys, os, bz2, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from compressjs_amd import synth

N = 100_000_000
def gen():
    np.save('/tmp/enwik.npy', synth.enwik_like(N, 2025))
    np.save('/tmp/text.npy', synth.text_like(N, 2025))
    for f in ('sample5.ref', 'sample3.ref'):
        p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle', '_ref', 'fixtures', f)
        if os.path.exists(p):
            d = np.fromfile(p, dtype=np.uint8)
            np.save('/tmp/%s.npy' % f.split('.')[0], np.tile(d, max(1, 50_000_000 // d.size)))
    np.save('/tmp/runs.npy', synth.runs_mixed(50_000_000, 3))

def run(names):
    import torch
    from compressjs_amd.bzip2 import Context
    ctx = Context(0, 128)
    tag = 'iters=%s tile=%s' % (os.environ.get('CJS_DEEP_ITERS', 'dflt'), os.environ.get('CJS_DEEP_TILE', 'dflt'))
    for name in names:
        p = '/tmp/%s.npy' % name
        if not os.path.exists(p):
            continue
        data = np.load(p)
        d_in = torch.from_numpy(data).cuda()
        cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
        d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
        t = []
        for _ in range(4):
            n = ctx.compress_device(d_in, d_out, 9)
            t.append(ctx.last_device_ms)
        out = d_out[:n].cpu().numpy().tobytes()
        rt = ''
        if '--check' in sys.argv:
            try:
                rt = ' roundtrip %s' % (bz2.decompress(out) == data.tobytes())
            except Exception as e:
                rt = ' DECODE ERROR %r' % (e,)
        print('[%s] %-8s %9d -> %9d  %7.2f ms  %7.1f MB/s  rounds %d sparse %d  sha %s%s' % (
            tag, name, data.size, n, min(t), data.size / min(t) / 1e3, ctx.L.cjs_dbg_k1_rounds(),
            ctx.L.cjs_dbg_k1_sparse_rounds(), hashlib.sha256(out).hexdigest()[:16], rt), flush=True)

if __name__ == '__main__':
    if sys.argv[1] == 'gen':
        gen()
    else:
        run([a for a in sys.argv[2:] if not a.startswith('--')])
