"""Timing sanity over different data shapes (not a test): python tests/gpu_perf_probe.py
Every row carries the digest of the GPU's stream next to the digest of the ORACLE's stream for the same bytes (oracle/bz2_oracle.c,
run for all shapes at once on the host's cores); libbz2's verdict is only a third column: a block that fills on the 4th byte of a
run gets no count byte from the reference (lib/Bzip2.js:640-644, mirrored on purpose), and libbz2 rejects such streams."""
import sys, os, time, bz2, hashlib
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

N = 50_000_000

def shapes(N=N):
    from compressjs_amd import synth
    rng = np.random.RandomState(1)
    _ = rng.randint(0, 256, size=N)
    out = [('two-symbol random', rng.randint(97, 99, size=N).astype(np.uint8)),
           ('text_like', synth.text_like(N, 2025)), ('lcg_ascii', synth.lcg_ascii(N, 7)), ('runs_mixed', synth.runs_mixed(N, 3)),
           ('periodic ab', synth.periodic(N, b'ab')), ('periodic 44B', synth.periodic(N, b'the quick brown fox jumps over the lazy dog\n')),
           ('zeros', np.zeros(N, np.uint8))]
    rng = np.random.RandomState(1)
    out.append(('random bytes', rng.randint(0, 256, size=N).astype(np.uint8)))
    base = synth.text_like(200_000, 5)
    out.append(('200k text tiled', np.tile(base, N // base.size)))
    for f in ('sample5.ref', 'sample4.ref', 'sample3.ref', 'sample2.ref'):
        p = os.path.join(ROOT, 'oracle', '_ref', 'fixtures', f)
        if os.path.exists(p):
            d = np.fromfile(p, dtype=np.uint8)
            out.append((f + ' tiled', np.tile(d, max(1, N // d.size))))
    return out

def oracle_digest(path):
    import oracle
    d = np.load(path)
    o = oracle.bz2_compress(d, 9)
    return hashlib.sha256(o).hexdigest()[:12], len(o)

def main():
    import torch
    from compressjs_amd.bzip2 import Context
    S = shapes()
    paths = []
    for i, (name, d) in enumerate(S):
        p = '/tmp/shape_%d.npy' % i
        np.save(p, d)
        paths.append(p)
    pool = ProcessPoolExecutor(max_workers=min(len(S), os.cpu_count() or 4))
    futs = [pool.submit(oracle_digest, p) for p in paths]
    ctx = Context(0, 128)
    rows = []
    for name, data in S:
        d_in = torch.from_numpy(data).cuda()
        cap = int(ctx.L.cjs_bz2_compress_bound(data.size))
        d_out = torch.zeros((cap + 3) & ~3, dtype=torch.uint8, device='cuda')
        n = ctx.compress_device(d_in, d_out, 9)
        t, hs = [], set()
        for _ in range(4):
            n = ctx.compress_device(d_in, d_out, 9); t.append(ctx.last_device_ms)
            hs.add(hashlib.sha256(d_out[:n].cpu().numpy().tobytes()).hexdigest()[:12])
        comp = d_out[:n].cpu().numpy().tobytes()
        try:
            lib = 'ok' if bz2.decompress(comp) == data.tobytes() else 'MISMATCH'
        except Exception as e:
            lib = 'rejects (%s)' % type(e).__name__
        back = torch.empty(data.size + 64, dtype=torch.uint8, device='cuda')
        nb = ctx.decompress_device(d_out[:n], back)
        own = bool(nb == data.size and torch.equal(back[:data.size], d_in))
        rows.append((name, data.size, n, min(t), ctx.last_block_count, ctx.L.cjs_dbg_k1_rounds(), ctx.L.cjs_dbg_k1_sparse_rounds(), sorted(hs), lib, own))
    for (name, size, n, ms, blocks, rounds, sparse, hs, lib, own), f in zip(rows, futs):
        od, ol = f.result()
        print('%-22s %9d -> %9d  %8.2f ms  %8.1f MB/s  blocks %3d rounds %2d sparse %2d  gpu %s  oracle %s  %s  GPU-decoder round trip %s  libbz2 %s' % (
            name, size, n, ms, size / ms / 1e3, blocks, rounds, sparse, '/'.join(hs), od,
            'EQUAL' if hs == [od] and n == ol else 'DIFFERENT', own, lib), flush=True)

if __name__ == '__main__':
    main()
