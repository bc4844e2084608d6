"""More of test_compress_differential_fuzz_vs_oracle with other seeds (not collected by pytest): python tests/gpu_fuzz_streams.py [seed] [cases]
Mixtures of runs / text / noise / periodic pieces at lengths around the block capacities and at random lengths, random levels, host-buffer path
(Context.compress) and device path alternately; the GPU stream must equal the oracle's bit for bit."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, oracle
from compressjs_amd import synth
from compressjs_amd.bzip2 import Context
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.RandomState(seed)
ctx = Context(0, 16)


def piece(n):
    k = rng.randint(0, 7)
    if k == 0:
        return rng.randint(0, 256, size=n).astype(np.uint8)
    if k == 1:
        return synth.text_like(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]
    if k == 2:
        return synth.runs_mixed(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]
    if k == 3:
        return np.full(n, rng.randint(0, 256), dtype=np.uint8)
    if k == 4:
        p = rng.randint(0, 256, size=rng.randint(1, 40)).astype(np.uint8)
        return np.tile(p, n // p.size + 1)[:n]
    if k == 5:
        return rng.randint(97, 97 + rng.randint(1, 5), size=n).astype(np.uint8)      # alphabets of 1 .. 4 symbols (fewer symbols than tables)
    return synth.lcg_ascii(max(n, 1), int(rng.randint(1, 1 << 20)))[:n]


for case in range(cases):
    level = int(rng.randint(1, 10))
    cap = level * 100000 - 19
    total = int(rng.choice([cap - 3, cap, cap + 1, cap + 255, 2 * cap + 17, 3 * cap - 1, 30011, 777, 1, 0, 1234567, int(rng.randint(0, 1_400_000)), int(rng.randint(0, 60_000))]))
    total = min(total, 1_400_000)
    parts, left = [], total
    while left > 0:
        n = int(min(left, rng.choice([1, 3, 4, 5, 255, 256, 1000, 4000, 50000, 99981, 300000])))
        parts.append(piece(n))
        left -= n
    d = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    want = oracle.bz2_compress(d, level)
    if case & 1 or d.size == 0:
        got = ctx.compress(d, level)
    else:
        d_in = torch.from_numpy(d).cuda()
        d_out = torch.zeros((int(ctx.L.cjs_bz2_compress_bound(d.size)) + 3) & ~3, dtype=torch.uint8, device='cuda')
        n = ctx.compress_device(d_in, d_out, level)
        got = d_out[:n].cpu().numpy().tobytes()
    assert got == want, (seed, case, level, total)
print('ok', seed, cases)
