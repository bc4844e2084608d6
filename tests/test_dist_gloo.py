"""world_size-4 test of the multi-GPU sharding logic (compressjs_amd/dist.py) on CPU: gloo
backend, kernels through the CPU logic-debug build.  The assembled stream must equal the
single-process stream bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stagelib
    from compressjs_amd import _lib, synth
    from compressjs_amd.bzip2 import Context
    from compressjs_amd.dist import sharded_compress
    _lib._lib = _lib.load(stagelib.EMU_SO)          # CPU logic-debug build stands in for the GPU
    ctx = Context(0, 2)
    data = np.concatenate([synth.text_like(230000, 21), synth.runs_mixed(40000, 2)])
    d_in = torch.from_numpy(data.copy())
    out = sharded_compress(ctx, d_in, 1)
    out = torch.from_numpy(np.frombuffer(out.numpy().tobytes(), np.uint8).copy()) if rank == 0 else out   # (a snapshot: ADVICE r4)
    # the sliced driver: every rank holds its slice + margin only; then an input whose blocks swallow slices (fallback)
    from compressjs_amd.dist import margin_bytes, sharded_compress_sliced, slice_bounds
    outs = []
    for dd in (data, np.concatenate([synth.lcg_ascii(150000, 4), np.zeros(900000, np.uint8)])):
        lo, hi = slice_bounds(dd.size, rank, world)
        wlo = max(0, lo - margin_bytes(1))
        o2 = sharded_compress_sliced(ctx, torch.from_numpy(dd[wlo:hi].copy()), wlo, dd.size, 1, d_all=lambda: torch.from_numpy(dd.copy()))
        outs.append(o2.numpy().tobytes() if rank == 0 else None)
    # the parallel plan (round 3): slice + FOLLOWING margin, one all_gather of summaries, no chain; inputs: text (clean
    # boundaries), a run that straddles a slice boundary, a boundary inside a long run (falls back to the sliced driver)
    from compressjs_amd.dist import sharded_compress_parallel
    par = []
    straddle = np.concatenate([synth.text_like(270000 // world - 2, 5), np.full(9, 65, np.uint8), synth.text_like(270000, 6)])[:270000]
    for dd in (data, straddle, np.concatenate([synth.lcg_ascii(150000, 4), np.zeros(900000, np.uint8)])):
        lo, hi = slice_bounds(dd.size, rank, world)
        whi = min(dd.size, hi + margin_bytes(1))

        def fb(dd=dd, lo=lo, hi=hi):
            wlo = max(0, lo - margin_bytes(1))
            return sharded_compress_sliced(ctx, torch.from_numpy(dd[wlo:hi].copy()), wlo, dd.size, 1, d_all=lambda: torch.from_numpy(dd.copy()))
        o3 = sharded_compress_parallel(ctx, torch.from_numpy(dd[lo:whi].copy()), hi - lo, lo, dd.size, 1, fallback=fb)
        par.append(o3.numpy().tobytes() if rank == 0 else None)
    # the same stream with the segments gathered by all_gather instead of grouped send/recv (CJS_DIST_GATHER: a switch for a first
    # contact with a fabric on which point-to-point misbehaves)
    clean = synth.lcg_ascii(470000, 6)                       # (plans slice by slice on four ranks: no fall-back involved)
    lo, hi = slice_bounds(clean.size, rank, world)
    whi = min(clean.size, hi + margin_bytes(1))
    o4a = sharded_compress_parallel(ctx, torch.from_numpy(clean[lo:whi].copy()), hi - lo, lo, clean.size, 1)
    o4a = o4a.numpy().tobytes() if rank == 0 else None
    os.environ["CJS_DIST_GATHER"] = "allgather"
    o4 = sharded_compress_parallel(ctx, torch.from_numpy(clean[lo:whi].copy()), hi - lo, lo, clean.size, 1)
    del os.environ["CJS_DIST_GATHER"]
    # round 6: a block boundary INSIDE a run of four or more equal bytes (ordinary text has them) no longer sends the job to the
    # replicated plan - the chain carries the shift: runs of 6 and 9 bytes planted on the serial plan's boundaries 1 and 3 (slices 0 and
    # 2 of four), planned in parallel, no fall-back allowed
    carried = synth.text_like(470000, 31).copy()
    nser = ctx.plan(torch.from_numpy(carried), 1)
    bnd = [ctx.plan_block_start(k) for k in range(nser)]
    carried[bnd[1] - 3:bnd[1] + 3] = 66
    carried[bnd[3] - 2:bnd[3] + 7] = 67
    lo, hi = slice_bounds(carried.size, rank, world)
    whi = min(carried.size, hi + margin_bytes(1))
    fell = []

    def no_fb():
        fell.append(1)
        return sharded_compress(ctx, torch.from_numpy(carried.copy()), 1)
    o5 = sharded_compress_parallel(ctx, torch.from_numpy(carried[lo:whi].copy()), hi - lo, lo, carried.size, 1, fallback=no_fb)
    assert not fell, "the chained plan fell back on a boundary inside a short run"
    o5 = o5.numpy().tobytes() if rank == 0 else None
    # ... and the fall-back path itself on an input that plans fine (a parameter of the call: ADVICE r5, no process-wide switch)
    o6 = sharded_compress_parallel(ctx, torch.from_numpy(carried[lo:whi].copy()), hi - lo, lo, carried.size, 1, fallback=no_fb, force_fallback=True)
    assert fell
    if rank == 0:
        assert o4.numpy().tobytes() == o4a
        assert o6.numpy().tobytes() == o5
        q.put((out.numpy().tobytes(), outs, par, o4a, o5, carried.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stream_equals_reference_stream():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stagelib
    stagelib.build_emu()
    import oracle
    from compressjs_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    world = 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, sliced, par, clean_stream, carried_stream, carried_in = q.get(timeout=1800)
    for p in procs:
        p.join(timeout=1800)
        assert p.exitcode == 0
    data = np.concatenate([synth.text_like(230000, 21), synth.runs_mixed(40000, 2)])
    assert got == oracle.bz2_compress(data, 1)
    assert sliced[0] == oracle.bz2_compress(data, 1)                                   # chained planning over three slices
    runs = np.concatenate([synth.lcg_ascii(150000, 4), np.zeros(900000, np.uint8)])
    assert sliced[1] == oracle.bz2_compress(runs, 1)                                   # a block swallows a slice: replicated fallback
    # parallel plan over four slices: text, a 9-byte run across a slice boundary, and the run-heavy input (falls back)
    assert par[0] == oracle.bz2_compress(data, 1)
    straddle = np.concatenate([synth.text_like(270000 // world - 2, 5), np.full(9, 65, np.uint8), synth.text_like(270000, 6)])[:270000]
    assert par[1] == oracle.bz2_compress(straddle, 1)
    assert par[2] == oracle.bz2_compress(runs, 1)
    assert clean_stream == oracle.bz2_compress(synth.lcg_ascii(470000, 6), 1)          # the parallel plan itself (no fall-back), both gathers
    cin = np.frombuffer(carried_in, np.uint8)
    assert (cin == 66).sum() >= 6 and carried_stream == oracle.bz2_compress(cin, 1)     # boundaries inside runs of 6 and 9: carried, not refused


def test_shift_and_trailer_helpers():
    from compressjs_amd.dist import shift_bits, trailer_bytes, _rotl32
    seg = torch.tensor([0b10110011, 0b01010101], dtype=torch.uint8)
    assert shift_bits(seg, 2, 0).tolist() == [0b10110011, 0b01010101, 0]
    assert shift_bits(seg, 2, 3).tolist() == [0b00010110, 0b01101010, 0b10100000]
    off, tb, total = trailer_bytes(32, 0)
    assert off == 4 and total == 14 and tb == bytes.fromhex("17724538509000000000")
    assert _rotl32(0x80000001, 1) == 0x00000003


def test_shift_kernel_equals_spec():
    """cjs_shift_bits (k5_shift_bits through the CPU debug build) against the torch restatement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stagelib
    from compressjs_amd import _lib
    from compressjs_amd.bzip2 import Context
    from compressjs_amd.dist import shift_bits
    stagelib.build_emu()
    saved = _lib._lib
    _lib._lib = _lib.load(stagelib.EMU_SO)
    try:
        ctx = Context(0, 2)
        g = torch.Generator().manual_seed(5)
        seg = torch.randint(0, 256, (5000,), dtype=torch.uint8, generator=g)
        for n in (0, 1, 2, 255, 256, 257, 4999):
            for s in range(8):
                out = torch.full((n + 1,), 0xAA, dtype=torch.uint8)
                ctx.shift_bits(seg, n, s, out)
                assert torch.equal(out, shift_bits(seg, n, s)), (n, s)
        ctx.close()
    finally:
        _lib._lib = saved


def test_parallel_plan_equals_serial_plan():
    """The parallel plan of compressjs_amd/dist.py (plan_bases + cjs_bz2_plan_phase) against the serial chain (cjs_bz2_plan) on
    one process: random inputs with runs placed on and around the cuts, random cuts.  Wherever every slice accepts, the union of
    the slices' block starts must be exactly the serial plan's; a slice may only refuse (fall back), never disagree."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stagelib
    from compressjs_amd import _lib, synth
    from compressjs_amd.bzip2 import Context
    from compressjs_amd.dist import _edge_runs, margin_bytes, plan_bases
    stagelib.build_emu()
    saved = _lib._lib
    _lib._lib = _lib.load(stagelib.EMU_SO)
    try:
        ctx = Context(0, 2)
        rng = np.random.RandomState(11)
        level, accepted, refused = 1, 0, 0
        for case in range(14):
            n = int(rng.randint(250_000, 420_000))
            d = synth.text_like(n, 100 + case) if case % 3 else synth.lcg_ascii(n, case + 1)
            d = d.copy()
            world = int(rng.choice([2, 3, 4, 5]))
            cuts = sorted(int(x) for x in rng.choice(np.arange(20_000, n - 20_000), size=world - 1, replace=False))
            for c in cuts:                                             # runs of 1..9 equal bytes ending before / on / after the cut
                L, sh = int(rng.randint(1, 10)), int(rng.randint(-9, 10))
                a = max(0, c + sh - L // 2)
                d[a:a + L] = 66
            if case % 5 == 4:
                d[cuts[0] - 700:cuts[0] + 900] = 0                     # a long run across a cut: must refuse
            t = torch.from_numpy(d)
            nser = ctx.plan(t, level)
            serial = [ctx.plan_block_start(k) for k in range(nser)]
            lo_hi = list(zip([0] + cuts, cuts + [n]))
            meta = []
            for lo, hi in lo_hi:
                w = t[lo:min(n, hi + margin_bytes(level))]
                ctx.plan_scan(w, level)
                meta.append((hi - lo, ctx.plan_cost(hi - lo)) + _edge_runs(w[:hi - lo]))
            bases = plan_bases(meta, level)
            got, ok_all = [], True
            for (lo, hi), (phase, ok) in zip(lo_hi, bases):
                w = t[lo:min(n, hi + margin_bytes(level))]
                ctx.plan_scan(w, level)
                nb = ctx.plan_phase(hi - lo, phase, hi == n) if ok else -1
                if nb < 0:
                    ok_all = False
                    break
                got += [lo + ctx.plan_block_start(k) for k in range(nb)]
            if ok_all:
                assert got == serial, (case, world, cuts)
                accepted += 1
            else:
                refused += 1
        assert accepted >= 8 and refused >= 1, (accepted, refused)
        # a block boundary exactly on (and around) the first byte of a run that straddles a cut: the serial plan
        # is [0, 99981, ...]; the slice that starts inside the run must refuse or agree, never add a block at its own start
        cap1 = 99981
        swept = 0
        for pre in range(cap1 - 3, cap1 + 4):
            for runlen, into in ((10, 5), (10, 1), (4, 2), (7, 6), (300, 5)):
                d = np.concatenate([synth.lcg_ascii(pre, 3), np.full(runlen, 65, np.uint8), synth.lcg_ascii(cap1 - 2, 4)])
                d[pre - 1], d[pre + runlen] = 66, 67                   # the run is exactly runlen long
                cut = pre + into
                n = d.size
                t = torch.from_numpy(d)
                nser = ctx.plan(t, level)
                serial = [ctx.plan_block_start(k) for k in range(nser)]
                meta = []
                for lo, hi in ((0, cut), (cut, n)):
                    w = t[lo:min(n, hi + margin_bytes(level))]
                    ctx.plan_scan(w, level)
                    meta.append((hi - lo, ctx.plan_cost(hi - lo)) + _edge_runs(w[:hi - lo]))
                got, ok_all = [], True
                for (lo, hi), (phase, ok) in zip(((0, cut), (cut, n)), plan_bases(meta, level)):
                    w = t[lo:min(n, hi + margin_bytes(level))]
                    ctx.plan_scan(w, level)
                    nb = ctx.plan_phase(hi - lo, phase, hi == n) if ok else -1
                    if nb < 0:
                        ok_all = False
                        break
                    got += [lo + ctx.plan_block_start(k) for k in range(nb)]
                if ok_all:
                    assert got == serial, (pre, runlen, into, got, serial)
                swept += 1
        assert swept == 35
        ctx.close()
    finally:
        _lib._lib = saved


def _chained_plan(ctx, t, lo_hi, level):
    """The protocol of sharded_compress_parallel on one process: every slice plans from its assumed target, chain_step validates,
    wrong assumptions are planned again.  Returns (block starts, passes) or (None, passes) when a slice refuses."""
    from compressjs_amd.dist import _edge_runs, chain_step, margin_bytes, plan_origins
    n = t.numel()
    cap = level * 100000 - 19
    wins = [t[lo:min(n, hi + margin_bytes(level))] for lo, hi in lo_hi]
    meta = []
    for (lo, hi), w in zip(lo_hi, wins):
        ctx.plan_scan(w, level)
        meta.append((hi - lo, ctx.plan_cost(hi - lo)) + _edge_runs(w[:hi - lo]))
    origins = plan_origins(meta, level)
    assumed = [(-o[0]) % cap for o in origins]
    planned = [None] * len(lo_hi)
    res = [None] * len(lo_hi)
    starts = [None] * len(lo_hi)
    for it in range(len(lo_hi) + 2):
        for r, ((lo, hi), w) in enumerate(zip(lo_hi, wins)):
            if planned[r] != assumed[r]:
                planned[r] = assumed[r]
                ctx.plan_scan(w, level)
                nb, tn = ctx.plan_chain(hi - lo, assumed[r], w.numel() == n - lo)
                res[r] = (tn, nb >= 0)
                starts[r] = [lo + ctx.plan_block_start(k) for k in range(nb)] if nb >= 0 else None
        status, assumed = chain_step(origins, list(planned), res, cap, ends=max([r for r, m in enumerate(meta) if m[0] > 0], default=None))
        if status == "done":
            return [s for st in starts for s in st], it + 1
        if status == "refuse":
            return None, it + 1
    return None, -1


def test_chained_plan_carries_boundaries_inside_runs():
    """Round 6: a block boundary inside a run of four or more equal bytes restarts the run in the new block and
    moves every later boundary - the chained plan (cjs_bz2_plan_chain + dist.chain_step) carries that shift from slice to slice
    instead of refusing.  Text inputs with runs of 4..9 (and 30, 300) bytes planted ON the serial plan's block boundaries, random
    cuts into 2..5 slices: the union of the slices' block starts must equal the serial chain's (cjs_bz2_plan) - and such inputs must
    be ACCEPTED (planned in parallel), needing more than one pass."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stagelib
    from compressjs_amd import _lib, synth
    from compressjs_amd.bzip2 import Context
    stagelib.build_emu()
    saved = _lib._lib
    _lib._lib = _lib.load(stagelib.EMU_SO)
    try:
        ctx = Context(0, 2)
        rng = np.random.RandomState(23)
        level = 1
        accepted = multi = refused = 0
        for case in range(16):
            n = int(rng.randint(420_000, 650_000))
            d = (synth.text_like(n, 300 + case) if case % 2 else synth.lcg_ascii(n, case + 7)).copy()
            t = torch.from_numpy(d)
            nser = ctx.plan(t, level)
            b0 = [ctx.plan_block_start(k) for k in range(nser)]
            # runs on one to three of the plan's own boundaries (the later ones move once the first is cut: the run is long enough to
            # cover the moved boundary more often than not)
            for k in rng.choice(np.arange(1, max(2, nser // 2 + 1)), size=min(int(rng.randint(1, 3)), max(1, nser // 2)), replace=False):
                L = int(rng.choice([4, 5, 6, 9, 30, 300]))
                a = b0[k] - int(rng.randint(1, L))
                d[a:a + L] = 66
            nser = ctx.plan(t, level)
            serial = [ctx.plan_block_start(k) for k in range(nser)]
            inrun = sum(1 for s in serial[1:] if d[s - 1] == d[s] and (d[s - 2] == d[s] or d[s + 1] == d[s]))
            world = int(rng.choice([2, 3, 4, 5]))
            # (the cuts behind the planted runs, mostly: a moved boundary only shows in the slices behind it)
            cuts = sorted(int(x) for x in rng.choice(np.arange(n // 2 if case % 4 else 30_000, n - 30_000), size=world - 1, replace=False))
            lo_hi = list(zip([0] + cuts, cuts + [n]))
            got, passes = _chained_plan(ctx, t, lo_hi, level)
            if got is None:
                refused += 1
                continue
            assert got == serial, (case, world, cuts, passes)
            accepted += 1
            multi += 1 if (passes > 1 and inrun) else 0
        assert accepted >= 14 and multi >= 5, (accepted, multi, refused)
        # the round-3 sweep again, chained: a boundary on and around the first byte of a run that straddles a cut - refuse or agree
        cap1 = 99981
        agreed = 0
        for pre in range(cap1 - 3, cap1 + 4):
            for runlen, into in ((10, 5), (10, 1), (4, 2), (7, 6), (300, 5)):
                d = np.concatenate([synth.lcg_ascii(pre, 3), np.full(runlen, 65, np.uint8), synth.lcg_ascii(cap1 - 2, 4)])
                d[pre - 1], d[pre + runlen] = 66, 67
                t = torch.from_numpy(d)
                nser = ctx.plan(t, level)
                serial = [ctx.plan_block_start(k) for k in range(nser)]
                cut = pre + into
                got, _ = _chained_plan(ctx, t, [(0, cut), (cut, d.size)], level)
                if got is not None:
                    assert got == serial, (pre, runlen, into, got, serial)
                    agreed += 1
        assert agreed >= 10, agreed
        ctx.close()
    finally:
        _lib._lib = saved
