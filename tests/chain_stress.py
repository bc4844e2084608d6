"""Stress of the chained slice plan on the CPU logic-debug build (not collected by pytest): random text / random / run-heavy inputs with runs of 2..1000 bytes planted
on the serial plan's block boundaries and on the cuts, cuts ON block boundaries, 2..8 slices: the union of the slices' plans must equal the serial chain's (cjs_bz2_plan) or
the slices must refuse.  python tests/chain_stress.py [seed] [cases]   (round 6: seed 99, 100 cases: 87 accepted and equal, 13 refused - run-heavy inputs -, 14 needed more than one pass)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, stagelib
from compressjs_amd import _lib, synth
from compressjs_amd.bzip2 import Context
import test_dist_gloo as T
_lib._lib = _lib.load(stagelib.EMU_SO)
ctx = Context(0, 2)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
level = 1
acc = ref = multi = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 120):
    n = int(rng.randint(250_000, 700_000))
    kind = case % 4
    d = (synth.text_like(n, 1000 + case) if kind < 2 else synth.lcg_ascii(n, case + 3) if kind == 2 else synth.runs_mixed(n, case)).copy()
    t = torch.from_numpy(d)
    nser = ctx.plan(t, level)
    b0 = [ctx.plan_block_start(k) for k in range(nser)]
    world = int(rng.choice([2, 3, 4, 5, 8]))
    cuts = sorted(set(int(x) for x in rng.choice(np.arange(20_000, n - 20_000), size=world - 1, replace=False)))
    # runs on boundaries, on cuts, and cuts ON boundaries
    for k in range(1, nser):
        if rng.rand() < 0.6:
            L = int(rng.choice([2, 3, 4, 5, 6, 9, 30, 254, 255, 256, 300, 1000]))
            a = max(0, b0[k] - int(rng.randint(0, L + 1)))
            d[a:a + L] = int(rng.choice([65, 66]))
    for i, c in enumerate(cuts):
        r = rng.rand()
        if r < 0.3:
            L = int(rng.choice([2, 4, 7, 40])); a = max(0, c - int(rng.randint(0, L + 1))); d[a:a + L] = 67
        elif r < 0.5 and nser > 2:
            cuts[i] = b0[int(rng.randint(1, nser))] + int(rng.randint(-2, 3))
    cuts = sorted(set(c for c in cuts if 1000 < c < n - 1000))
    nser = ctx.plan(t, level)
    serial = [ctx.plan_block_start(k) for k in range(nser)]
    lo_hi = list(zip([0] + cuts, cuts + [n]))
    got, passes = T._chained_plan(ctx, t, lo_hi, level)
    if got is None:
        ref += 1
    else:
        assert got == serial, (case, n, cuts, got, serial)
        acc += 1; multi += passes > 1
print('accepted', acc, 'refused', ref, 'multi-pass', multi)
