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
    if world > 1:                                                 # grouped send/recv, all peers' links into the root used at once)
        root = dist.get_global_rank(group, 0) if group is not None else 0
        if rank == 0:
            gl = [shifted] + [torch.empty(lens[r], dtype=torch.uint8, device=cdev) for r in range(1, world)]
            ops = [dist.P2POp(dist.irecv, gl[r], dist.get_global_rank(group, r) if group is not None else r, group=group)
                   for r in range(1, world)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            gl = [t.to(dev) for t in gl]
        else:
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, shifted.to(cdev), root, group=group)]):
                req.wait()
            gl = None
    else:
        gl = [shifted]
    mark("gather")
    if rank != 0:
        return None
    toff, tbytes, total = trailer_bytes(pos, crc)
    final = torch.zeros(total + 8, dtype=torch.uint8, device=dev)
    final[:4] = torch.tensor(list(b"BZh" + bytes([48 + level])), dtype=torch.uint8, device=dev)
    for r, (b, _, _) in enumerate(meta):
        n = (b + 7) // 8 + 1
        o = offs[r] // 8
        n = min(n, final.numel() - o)
        final[o:o + n] |= gl[r][:n]
    tb = torch.tensor(list(tbytes), dtype=torch.uint8, device=dev)
    final[toff:toff + tb.numel()] |= tb
    mark("assemble")
    if trace is not None:
        print("[dist] " + ", ".join("%s %.2f ms" % (n, (t - trace[i][1]) * 1e3) for i, (n, t) in enumerate(trace[1:])), flush=True)
    return final[:total]
