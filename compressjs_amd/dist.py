"""Multi-GPU sharding of the bzip2 block pipeline (one process per GPU, torch.distributed).

bzip2 blocks are independent once the input-dependent RLE1 split is known, so:

  1. every rank runs the cheap split pre-pass (cjs_bz2_plan, K0) over the whole input it holds
     -- no data-path collective is needed to agree on block boundaries;
  2. rank r encodes the contiguous block range [r*nb/W, (r+1)*nb/W) into a bit stream that
     starts at bit 0 (cjs_bz2_encode_blocks);
  3. one all_gather of (bits, crc_fold, block count) per rank -- 24 bytes each -- gives every
     rank its absolute bit offset; each rank shifts its own segment by (offset mod 8) bits;
  4. the shifted segments go to rank 0 at their own lengths - grouped send/recv (RCCL has no
     gatherv), all peers' xGMI links into the root in use at once; payload = compressed bytes
     only - which ORs them together at their byte offsets, writes "BZh<level>" and the
     end-of-stream magic + combined CRC (lib/Bzip2.js:903-906, 917, 925-927).

The combined CRC is linear over GF(2): S' = rotl^k(S) ^ P with P the shard's fold, so shards
chain without exchanging per-block CRCs.  Everything below is plain torch ops and works on CPU
tensors too (the gloo tests drive it with the CPU logic-debug build of the kernels)."""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist


def _rotl32(v: int, k: int) -> int:
    k %= 32
    v &= 0xFFFFFFFF
    return ((v << k) | (v >> (32 - k))) & 0xFFFFFFFF if k else v


def shift_bits(seg: torch.Tensor, nbytes: int, s: int) -> torch.Tensor:
    """Return seg[:nbytes] shifted right by s (0..7) bits as nbytes+1 bytes (MSB-first stream)."""
    b = seg[:nbytes].to(torch.int16)
    out = torch.zeros(nbytes + 1, dtype=torch.int16, device=seg.device)
    if s == 0:
        out[:nbytes] = b
    else:
        out[:nbytes] = b >> s
        out[1:nbytes + 1] |= (b << (8 - s)) & 0xFF
    return out.to(torch.uint8)


def block_range(nblocks: int, rank: int, world: int):
    per = (nblocks + world - 1) // world
    first = min(rank * per, nblocks)
    return first, min(per, nblocks - first)


def trailer_bytes(bit_pos: int, crc: int):
    """End-of-stream magic (48 bits) + combined CRC (32 bits) placed at bit_pos; returns
    (byte offset, bytes, total stream bytes)."""
    val = (0x177245385090 << 32) | (crc & 0xFFFFFFFF)
    end = bit_pos + 80
    pad = (8 - end % 8) % 8
    first = bit_pos // 8
    nbytes = (end + pad) // 8 - first
    lead = bit_pos - first * 8
    v = val << (nbytes * 8 - lead - 80)
    return first, v.to_bytes(nbytes, "big"), (end + pad) // 8


def _same_device(a, b) -> bool:
    """torch.device('cuda') and torch.device('cuda:0') name the same device when 0 is the current one."""
    a, b = torch.device(a), torch.device(b)
    if a.type != b.type:
        return False
    if a.type != "cuda":
        return True
    cur = torch.cuda.current_device() if torch.cuda.is_available() else 0
    return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)


def _assemble(ctx, seg, bits, fold, cnt, level, group, rank, world, dev, mark):
    """Shared tail of both drivers: 24-byte all_gather -> bit offsets and CRC fold; shift own segment; variable-length
    send/recv to rank 0; rank 0 ORs the segments, writes header and trailer.  Returns the stream on rank 0, else None."""
    # collectives run on the tensors' own device with RCCL ("nccl"); with the gloo backend (CPU tests,
    # or several ranks sharing one GPU) they are staged through host memory
    cdev = dev if (world == 1 or dist.get_backend(group) != "gloo") else torch.device("cpu")
    mine = torch.tensor([bits, fold, cnt], dtype=torch.int64, device=cdev)
    if world > 1:
        allv = [torch.zeros(3, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(allv, mine, group=group)
        meta = torch.stack(allv).cpu().tolist()
    else:
        meta = [mine.cpu().tolist()]
    offs, pos, crc = [], 32, 0
    for b, f, k in meta:
        offs.append(pos)
        pos += b
        crc = _rotl32(crc, k) ^ (f & 0xFFFFFFFF)
    mark("all_gather")
    my_off = offs[rank]
    nbytes = (bits + 7) // 8
    shifted = torch.zeros(nbytes + 1, dtype=torch.uint8, device=dev)
    ctx.shift_bits(seg, nbytes, my_off % 8, shifted)              # one pass (k5_shift_bits); shift_bits() below is the spec
    mark("shift")
    lens = [(b + 7) // 8 + 1 for b, _, _ in meta]                  # every segment travels at its own length (RCCL has no gatherv:
    gl = None
    use_allgather = world > 1 and os.environ.get("CJS_DIST_GATHER", "p2p") == "allgather"
    if world > 1 and not use_allgather:                           # grouped send/recv, all peers' links into the root used at once)
        root = dist.get_global_rank(group, 0) if group is not None else 0
        if rank == 0:
            # one staging buffer for all peers' segments, kept across steps (no allocation inside the timed step)
            need = sum(lens[1:])
            stage = getattr(ctx, "_asm_stage", None)
            if stage is None or stage.numel() < need or not _same_device(stage.device, cdev):
                stage = torch.empty(need + (need >> 2) + 4096, dtype=torch.uint8, device=cdev)
                ctx._asm_stage = stage
            gl, o2 = [shifted], 0
            for r in range(1, world):
                gl.append(stage[o2:o2 + lens[r]])
                o2 += lens[r]
            ops = [dist.P2POp(dist.irecv, gl[r], dist.get_global_rank(group, r) if group is not None else r, group=group)
                   for r in range(1, world)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            gl = [t.to(dev) for t in gl]
        else:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, shifted.to(cdev), root, group=group)]):
                req.wait()
            gl = None
    elif use_allgather:
        # CJS_DIST_GATHER=allgather: the most ordinary collective instead of grouped point-to-point (every rank receives every
        # segment, padded to the longest: N times the traffic) - a switch for a first contact with a fabric on which the
        # send/recv path misbehaves; same bytes
        mx = max(lens)
        pad = torch.zeros(mx, dtype=torch.uint8, device=cdev)
        pad[:shifted.numel()] = shifted.to(cdev)
        allp = [torch.empty(mx, dtype=torch.uint8, device=cdev) for _ in range(world)]
        dist.all_gather(allp, pad, group=group)
        gl = [allp[r][:lens[r]].to(dev) for r in range(world)] if rank == 0 else None
    else:
        gl = [shifted]
    mark("gather")
    if rank != 0:
        return None
    toff, tbytes, total = trailer_bytes(pos, crc)
    final = getattr(ctx, "_asm_final", None)                  # the output buffer is kept across steps too
    if final is None or final.numel() < total + 8 or not _same_device(final.device, dev):
        final = torch.empty(total + (total >> 3) + 4096, dtype=torch.uint8, device=dev)
        ctx._asm_final = final
    final[:total + 8].zero_()
    final[:4] = torch.tensor(list(b"BZh" + bytes([48 + level])), dtype=torch.uint8, device=dev)
    for r, (b, _, _) in enumerate(meta):
        n = (b + 7) // 8 + 1
        o = offs[r] // 8
        n = min(n, total + 8 - o)
        final[o:o + n] |= gl[r][:n]
    tb = torch.tensor(list(tbytes), dtype=torch.uint8, device=dev)
    final[toff:toff + tb.numel()] |= tb
    mark("assemble")
    # a copy: `final` is scratch that the next call on this ctx zeroes and overwrites (a view of it changed under the caller)
    return final[:total].clone()


def sharded_compress(ctx, d_in: torch.Tensor, level: int, group=None, seg: torch.Tensor = None):
    """Compress d_in (the WHOLE stream, resident on every rank) with the blocks sharded over the
    ranks of `group`.  Returns the complete .bz2 stream as a uint8 tensor on rank 0 (None on
    the other ranks)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = d_in.device
    trace = [] if os.environ.get("CJS_DIST_TRACE") else None

    def mark(name):
        if trace is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            trace.append((name, time.perf_counter()))

    mark("start")
    nblocks = ctx.plan(d_in, level)
    mark("plan")
    first, count = block_range(nblocks, rank, world)
    if seg is None:
        per = (nblocks + world - 1) // world
        seg_cap = per * (level * 100000 * 2 + 32768) + 4096
        seg = torch.zeros(seg_cap, dtype=torch.uint8, device=dev)
    bits, fold, cnt = ctx.encode_blocks(first, count, seg)
    mark("encode")
    out = _assemble(ctx, seg, bits, fold, cnt, level, group, rank, world, dev, mark)
    if trace is not None:
        print("[dist] " + ", ".join("%s %.2f ms" % (n, (t - trace[i][1]) * 1e3) for i, (n, t) in enumerate(trace[1:])), flush=True)
    return out


def slice_bounds(total: int, rank: int, world: int):
    """Byte range [lo, hi) of the stream that rank `rank` holds in the sliced driver."""
    per = (total + world - 1) // world
    return min(rank * per, total), min((rank + 1) * per, total)


def margin_bytes(level: int) -> int:
    """Bytes of the previous rank's slice a rank needs in front of its own (its first block starts in there)."""
    return 4 * level * 100000


def sharded_compress_sliced(ctx, d_win: torch.Tensor, win_lo: int, total: int, level: int, group=None,
                            seg: torch.Tensor = None, d_all=None):
    """The same stream as sharded_compress, but every rank holds only ITS slice of the input ("each GPU:
    H2D its input slice"): d_win = bytes [win_lo, hi) of the stream with hi = slice_bounds(...)[1] and
    win_lo = max(0, lo - margin_bytes(level)) - the slice plus the tail of the previous rank's slice.

    Planning is chained through the ranks exactly as in cjs_bz2_compress_multi: rank r receives s_r, the first byte of
    its first block (8 bytes from rank r-1, sent as soon as THAT rank has planned, K0's pre-pass, not encoded), plans
    [s_r, hi) as an input of its own - a bzip2 block starts with a fresh RLE1 state (lib/Bzip2.js:636-667) -, sends
    s_(r+1) = the start of its last, incomplete block on, and encodes the others.  The rest (24-byte all_gather, seam
    shift, variable-length send/recv, assembly) is shared with sharded_compress.  When a block reaches further back than
    the margin or swallows a whole slice (run-heavy input), all ranks fall back to the replicated driver on `d_all`
    (callable returning the whole stream as a device tensor, or None: then RuntimeError)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = d_win.device
    trace = [] if os.environ.get("CJS_DIST_TRACE") else None

    def mark(name):
        if trace is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            trace.append((name, time.perf_counter()))

    mark("start")
    lo, hi = slice_bounds(total, rank, world)
    cdev = dev if (world == 1 or dist.get_backend(group) != "gloo") else torch.device("cpu")
    grank = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    BAD = -1
    s = 0
    if rank > 0:
        t = torch.zeros(1, dtype=torch.int64, device=cdev)
        dist.recv(t, src=grank(rank - 1), group=group)
        s = int(t.item())
    bad = s == BAD or (rank > 0 and (s < win_lo or s > hi))
    keep, nxt = 0, BAD
    if not bad:
        nb = ctx.plan(d_win[s - win_lo:], level) if hi > s else 0
        if hi < total:
            if nb < 2:
                bad = True                                   # one block swallowed the slice
            else:
                keep = nb - 1
                nxt = s + ctx.plan_block_start(keep)
        else:
            keep, nxt = nb, total
    if rank + 1 < world:
        dist.send(torch.tensor([BAD if bad else nxt], dtype=torch.int64, device=cdev), dst=grank(rank + 1), group=group)
    mark("plan chain")
    flag = torch.tensor([1 if bad else 0], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        if d_all is None:
            raise RuntimeError("sliced sharding does not apply to this input (a block reaches beyond the margin); use sharded_compress")
        return sharded_compress(ctx, d_all(), level, group=group)
    if seg is None:
        seg = torch.zeros(((hi - lo) + margin_bytes(level)) * 3 // 2 + (1 << 20), dtype=torch.uint8, device=dev)
    if keep:
        bits, fold, cnt = ctx.encode_blocks(0, keep, seg)
    else:
        bits, fold, cnt = 0, 0, 0
    mark("encode")
    out = _assemble(ctx, seg, bits, fold, cnt, level, group, rank, world, dev, mark)
    if trace is not None:
        print("[dist] " + ", ".join("%s %.2f ms" % (n, (t - trace[i][1]) * 1e3) for i, (n, t) in enumerate(trace[1:])), flush=True)
    return out


# ---------------------------------------------------------------------------------------------
# parallel plan (round 3): no rank waits for another rank's plan
# ---------------------------------------------------------------------------------------------
def _g(k: int) -> int:
    """RLE1 output bytes of the first k bytes of a fresh run (k0_g in csrc/k0_rle1.hip)."""
    q, r = divmod(k, 255)
    return 5 * q + (r if r < 4 else 5)


_EDGE = 4096          # bytes of each end of a slice that are looked at for its boundary runs; a longer run -> fall back


def _edge_runs(d_own: torch.Tensor):
    """(head byte, head run length, tail byte, tail run length, long) of a slice, from its first / last _EDGE bytes."""
    n = d_own.numel()
    if n == 0:
        return 0, 0, 0, 0, False
    head = d_own[:min(n, _EDGE)].cpu().numpy()
    tail = d_own[max(0, n - _EDGE):].cpu().numpy()
    hb, tb = int(head[0]), int(tail[-1])
    ne = (head != hb).nonzero()[0]
    lh = int(ne[0]) if ne.size else int(head.size)
    ne = (tail != tb).nonzero()[0]
    lt = int(tail.size - 1 - ne[-1]) if ne.size else int(tail.size)
    long_run = (lh == head.size and n > head.size) or (lt == tail.size and n > tail.size)
    return hb, lh, tb, lt, long_run


def plan_bases(meta, level: int):
    """From the gathered per-slice summaries [(own_len, cost, head byte, head run, tail byte, tail run, long)] to, per rank,
    (phase, ok): the RLE1 cost prefix G of the whole stream at the slice's first byte, corrected for a run that straddles
    the slice start, taken modulo the block capacity.  Pure integer arithmetic, identical on every rank."""
    cap = level * 100000 - 19
    out = []
    G = 0                       # cost prefix of the stream at the current slice start
    inb, ink = -1, 0            # the run that reaches the slice start from the left: byte, length so far
    for n, cost, hb, lh, tb, lt, long_run in meta:
        ok = not long_run
        delta = 0
        if n and inb == hb and ink > 0:
            delta = _g(ink + lh) - _g(ink) - _g(lh)          # the head run costs what the tail of a longer run costs
            if ink + lh >= 4:
                # a run of four or more bytes straddles the slice start: this rank's own prefix is wrong inside it, so no
                # block boundary may fall into its cost span - measured from the run's FIRST byte (in an earlier slice), not
                # from the slice start: a boundary on the run's first byte restarts the run in the new block, and the previous
                # rank plans exactly that block, while this rank (phase 0, no run check at its own first byte) would emit a
                # second block at its slice start (norun(99981) + 'A' * 10 + norun(99979) cut at 99986, level 1)
                c0, c1 = G - _g(ink), G + _g(ink + lh) - _g(ink)
                if c0 // cap != c1 // cap or c0 % cap == 0 or c1 % cap == 0:
                    ok = False
        base = G + delta                                     # G(lo + i) = base + C_own(i) beyond the head run
        out.append(((-base) % cap, ok))
        true_cost = cost + delta
        if n:
            if lh == n and inb == hb and ink > 0:
                ink += n                                     # the whole slice continues the incoming run
            elif lh == n:
                inb, ink = hb, n
            else:
                inb, ink = tb, lt
        G += true_cost
    return out


def plan_origins(meta, level: int):
    """plan_bases for the CHAINED plan (round 6): per slice (base, span, long_run) - base = G(lo) + head-run correction, the origin
    of the slice's own cost prefix in the stream's (G(lo + i) = base + C_own(i) beyond the head run); span = (c0, c1), the cost span
    of a run of four or more bytes that straddles the slice start, measured from the run's first byte (the slice's own prefix is
    wrong inside it: a boundary target in there cannot be planned by this slice), or None."""
    out = []
    G = 0
    inb, ink = -1, 0
    for n, cost, hb, lh, tb, lt, long_run in meta:
        delta, span = 0, None
        if n and inb == hb and ink > 0:
            delta = _g(ink + lh) - _g(ink) - _g(lh)
            if ink + lh >= 4:
                span = (G - _g(ink), G + _g(ink + lh) - _g(ink))
        out.append((G + delta, span, bool(long_run)))
        if n:
            if lh == n and inb == hb and ink > 0:
                ink += n
            elif lh == n:
                inb, ink = hb, n
            else:
                inb, ink = tb, lt
        G += cost + delta
    return out


def chain_step(origins, assumed, results, cap: int, ends=None):
    """One validation pass over the slices' speculative plans - pure integer arithmetic on gathered values, identical on every rank.
    origins: plan_origins(); assumed[r]: the t0 slice r planned with; results[r] = (t_next, ok) of that plan; ends: index of the
    slice in which the stream ends (the walk stops there).
    Returns (status, assumed'): status 'done' (every slice planned from its true target), 'refuse' (a slice cannot be planned on
    its own: fall back), or 'again' with the targets to plan from next: the first slice whose assumption was wrong gets its true
    target, the slices behind it the same shift as a guess."""
    tau = 0                                   # the stream's first boundary: G = 0
    for r, (base, span, long_run) in enumerate(origins):
        if ends is not None and r > ends:
            break                             # (slices behind the end of the stream: nothing to plan, nothing to agree on)
        if long_run:
            return "refuse", assumed
        if span is not None and span[0] <= tau <= span[1]:
            return "refuse", assumed          # the boundary falls into a run that straddles this slice's start
        true_t0 = max(0, tau - base)
        if assumed[r] != true_t0:
            shift = true_t0 - assumed[r]
            nxt = list(assumed)
            nxt[r] = true_t0
            for q in range(r + 1, len(origins)):
                g = assumed[q] + shift
                nxt[q] = g if g >= 0 else g + cap
            return "again", nxt
        t_next, ok = results[r]
        if not ok:
            return "refuse", assumed
        tau = base + t_next
    return "done", assumed


def sharded_compress_parallel(ctx, d_win: torch.Tensor, own_len: int, lo: int, total: int, level: int, group=None,
                              seg: torch.Tensor = None, fallback=None, force_fallback: bool = False):
    """The stream of sharded_compress with every rank holding its slice and a margin of what FOLLOWS it - d_win = stream
    bytes [lo, min(total, lo + own_len + margin_bytes(level))) - and no chain between the ranks' plans (the reference's
    `do { readBlock } while`, lib/Bzip2.js:913-922, is serial; the round-2 driver kept it serial across ranks):

      1. every rank scans its own window (K0's cost prefix) and summarises its slice: cost total, first and last run;
      2. ONE all_gather (7 integers per rank) gives every rank the stream's cost prefix at its slice start, hence the target
         of the first block boundary inside its slice: it plans the blocks that START in its slice (cjs_bz2_plan_chain), the
         margin completing the last one;
      3. one all_gather of (target planned from, target handed on, ok) BEFORE anything is encoded: every rank walks the chain of
         targets (chain_step).  Round 6: a boundary inside a run of four or more equal bytes no longer sends the job to the
         replicated plan - it shifts the targets behind it, the ranks behind it plan again (one more small all_gather) and the job
         stays sharded; what still falls back to `fallback()` (a callable that runs one of the other drivers): a block longer than
         the margin, a run that fills a block, a boundary run longer than 4 KB at a slice edge, a boundary inside a run that
         straddles a cut, any local error;
      4. the blocks are encoded; one all_reduce of a flag carries local errors of the encoding (then all fall back); else the
         (bits, CRC fold, block count) all_gather of the other drivers follows and segments are shifted, sent and assembled as before.
    Returns the stream on rank 0, None elsewhere."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = d_win.device
    trace = [] if os.environ.get("CJS_DIST_TRACE") else None

    def mark(name):
        if trace is not None:
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            trace.append((name, time.perf_counter()))

    mark("start")
    cdev = dev if (world == 1 or dist.get_backend(group) != "gloo") else torch.device("cpu")
    ctx.plan_scan(d_win, level)
    cost = ctx.plan_cost(own_len)
    hb, lh, tb, lt, long_run = _edge_runs(d_win[:own_len])
    mark("scan")
    mine = torch.tensor([own_len, cost, hb, lh, tb, lt, 1 if long_run else 0], dtype=torch.int64, device=cdev)
    if world > 1:
        allv = [torch.zeros(7, dtype=torch.int64, device=cdev) for _ in range(world)]
        dist.all_gather(allv, mine, group=group)
        meta = [tuple(int(x) for x in v.cpu().tolist()) for v in allv]
    else:
        meta = [tuple(int(x) for x in mine.cpu().tolist())]
    mark("all_gather summaries")
    # ---- the chained plan (round 6): every rank plans from the target it EXPECTS its first boundary to have (no boundary in front of
    # it moved: the round-3 phase); one all_gather of (assumed target, target handed on, ok) lets every rank walk the chain - the same
    # integer arithmetic everywhere (chain_step).  A boundary inside a run of four or more equal bytes somewhere in the job moves the
    # targets behind it by a few bytes: the ranks behind it plan again from the corrected targets (one more small all_gather per such
    # boundary; ordinary text: one job in four at 8 x 10^8 bytes has one), instead of all ranks falling back to the replicated plan.
    cap = level * 100000 - 19
    origins = plan_origins([(m[0], m[1], m[2], m[3], m[4], m[5], bool(m[6])) for m in meta], level)
    assumed = [(-o[0]) % cap for o in origins]
    last = lo + d_win.numel() >= total                         # the WINDOW ends where the stream ends: the final block may be short (a slice-level flag refused jobs whose last block starts in the slice before the last)
    nb, t_next, okp, planned_with = -1, 0, False, None
    bits, fold, cnt = 0, 0, 0
    status = "again"
    for _it in range(world + 2):
        # a local failure of any kind becomes the fall-back flag: every rank must reach the collective below, or the others hang
        # (ADVICE r3); the fall-back driver then raises the same error on this rank where every rank sees it
        if planned_with != assumed[rank]:
            planned_with = assumed[rank]
            try:
                nb, t_next = ctx.plan_chain(own_len, planned_with, last)
                okp = nb >= 0
            except Exception:                                    # noqa: BLE001
                if os.environ.get('CJS_DIST_TRACE'):
                    import traceback
                    traceback.print_exc()
                nb, t_next, okp = -1, planned_with, False
        if force_fallback:                                       # (tests: the fall-back path on an input that plans fine)
            okp = False
        mine3 = torch.tensor([planned_with, t_next, 1 if okp else 0], dtype=torch.int64, device=cdev)
        if world > 1:
            all3 = [torch.zeros(3, dtype=torch.int64, device=cdev) for _ in range(world)]
            dist.all_gather(all3, mine3, group=group)
            res = [tuple(int(x) for x in v.cpu().tolist()) for v in all3]
        else:
            res = [tuple(int(x) for x in mine3.cpu().tolist())]
        status, assumed = chain_step(origins, [r[0] for r in res], [(r[1], bool(r[2])) for r in res], cap,
                                     ends=max([r for r, m in enumerate(meta) if m[0] > 0], default=None))
        if status != "again":
            break
    mark("plan")
    if status != "done":
        if fallback is None:
            raise RuntimeError("this input cannot be planned slice by slice (a block longer than the margin, a run that fills a block or reaches "
                               "beyond 4 KB at a slice edge, a boundary inside a run that straddles a cut); use sharded_compress_sliced / sharded_compress")
        return fallback()
    try:
        if nb > 0:
            if seg is None:
                seg = torch.zeros((own_len + margin_bytes(level)) * 3 // 2 + (1 << 20), dtype=torch.uint8, device=dev)
            bits, fold, cnt = ctx.encode_blocks(0, nb, seg)
    except Exception:                                        # noqa: BLE001
        if os.environ.get('CJS_DIST_TRACE'):
            import traceback
            traceback.print_exc()
        nb = -1
    bad = nb < 0
    if seg is None:
        seg = torch.zeros(1 << 12, dtype=torch.uint8, device=dev)
    mark("encode")
    # one small all_reduce carries the flag "could not plan on my own" (it precedes the assembly's all_gather: a rank that
    # falls back never enters _assemble)
    flag = torch.tensor([1 if bad else 0], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()):
        if fallback is None:
            raise RuntimeError("this input cannot be planned slice by slice (a block boundary inside a long run, or a block longer "
                               "than the margin); use sharded_compress_sliced / sharded_compress")
        return fallback()
    mark("flag")
    out = _assemble(ctx, seg, bits, fold, cnt, level, group, rank, world, dev, mark)
    if trace is not None:
        print("[dist r%d] " % rank + ", ".join("%s %.2f ms" % (n, (t - trace[i][1]) * 1e3) for i, (n, t) in enumerate(trace[1:])), flush=True)
    return out
