"""Round-2 timing probe (not a test): python tests/gpu_r2_probe.py gen | run <workload>... [--batch N] [--size N]
`gen` caches the streams of tests/workloads.py under /tmp; `run` compresses each 4x and prints the best
device time, K1 round counts and the stream digest (which must not depend on any tuning knob)."""
import sys, os, hashlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads

def arg(name, dflt):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt

N = arg('--size', 100_000_000)
def path(w): return '/tmp/wl_%s_%d.npy' % (w, N)

def gen(names):
    for w in names:
        if not os.path.exists(path(w)):
            t = time.time(); np.save(path(w), workloads.stream(w, N)); print('gen', w, '%.1fs' % (time.time() - t), flush=True)

def run(names):
    import torch
    from compressjs_amd.bzip2 import Context
    batch = arg('--batch', 128)
    ctx = Context(0, batch)
    tag = 'batch=%d %s' % (batch, ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('CJS_')))
    for name in names:
        if not os.path.exists(path(name)):
            gen([name])
        data = np.load(path(name))
        d_in = torch.from_numpy(data).cuda()
        cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
        d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
        t = []
        for _ in range(arg('--reps', 4)):
            n = ctx.compress_device(d_in, d_out, 9)
            t.append(ctx.last_device_ms)
        out = d_out[:n].cpu().numpy().tobytes()
        med = sorted(t)[len(t) // 2]
        print('[%s] %-6s %9d -> %9d  %7.2f ms (median %.2f)  %7.1f MB/s  rounds %d sparse %d  sha %s' % (
            tag, name, data.size, n, min(t), med, data.size / min(t) / 1e3, ctx.L.cjs_dbg_k1_rounds(),
            ctx.L.cjs_dbg_k1_sparse_rounds(), hashlib.sha256(out).hexdigest()[:16]), flush=True)

if __name__ == '__main__':
    names = [a for a in sys.argv[2:] if a in workloads.NAMES]
    (gen if sys.argv[1] == 'gen' else run)(names)
