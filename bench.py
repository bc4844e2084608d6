#!/usr/bin/env python3
"""bench.py -- bzip2 -9 compress throughput of the MI355X block pipeline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete bzip2 -9 compression (Bzip2.compressFile equivalent, lib/Bzip2.js:879)
of the synthetic enwik8-shaped stream (BASELINE.json configs[2]: 10^8 bytes per GPU, ~112
blocks of 899 981 bytes; compressjs_amd.synth.enwik_like: words + wiki markup + phrase reuse,
calibrated so that bzip2 -9 reaches enwik8's ratio 0.29), input resident in HBM when the timed region starts, complete .bz2
stream resident in HBM (rank 0) when it ends.  Weak scaling: N GPUs compress an N x 10^8-byte
stream; blocks are sharded, the encoded segments are gathered to rank 0 (compressjs_amd/dist.py).
Prints ONE JSON line (rank 0)."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def cpu_baseline(data: np.ndarray, level: int, sample_bytes: int):
    """The oracle (C port of the reference algorithm) timed on ONE host core on the first
    `sample_bytes` of the same stream.  Checker code: only this leg may call it."""
    import oracle
    sample = data[:sample_bytes]
    t0 = time.perf_counter()
    out = oracle.bz2_compress(sample, level)
    dt = time.perf_counter() - t0
    return dict(value=round(sample.size / dt / 1e6, 4), unit="MB/s", cores=1, kind="port",
                sample="first %d bytes of the same stream, 1 thread, oracle/bz2_oracle.c; "
                       "the reference itself (node 12, 1 thread) measured 0.378 MB/s on this "
                       "path in the build container (BASELINE.md)" % sample.size), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100_000_000, help="input bytes per GPU")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--workload", default="enwik", choices=["enwik", "text", "lcg", "e8sa"],
                    help="enwik (default): synthetic words+markup with phrase reuse calibrated to enwik8's bzip2 -9 ratio 0.29; "
                         "text: the same without phrase reuse (ratio 0.38, an easier suffix structure); lcg: random printable ASCII (configs[3]); "
                         "e8sa: test/sample5.ref || test/sample4.ref tiled (SURVEY.md 8d E8S-A; needs the staged fixtures)")
    ap.add_argument("--cpu-sample", type=int, default=12_000_000)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--batch", type=int, default=128, help="bzip2 blocks in flight (over all streams)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    ndev = torch.cuda.device_count()
    backend = os.environ.get("CJS_DIST_BACKEND", "nccl")    # "gloo": ranks may share a GPU (test rigs only)
    if backend == "nccl":
        assert local < ndev, "one process per GPU: LOCAL_RANK %d but %d GPUs visible" % (local, ndev)
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from compressjs_amd import synth
    from compressjs_amd.bzip2 import Context
    from compressjs_amd.dist import sharded_compress

    total = args.size * world
    if args.workload == "e8sa":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import cases
        parts = [cases.fixture_path("sample5.ref"), cases.fixture_path("sample4.ref")]
        assert all(parts), "e8sa needs test/sample5.ref and test/sample4.ref (staged by __graft_entry__.build())"
        base = np.concatenate([np.fromfile(p, dtype=np.uint8) for p in parts])
        host = np.tile(base, total // base.size + 1)[:total].copy()
    else:
        host = (synth.text_like(total, 2025) if args.workload == "text" else
                synth.enwik_like(total, 2025) if args.workload == "enwik" else synth.lcg_ascii(total, 7))
    d_in = torch.from_numpy(host).to(dev)
    ctx = Context(local, args.batch)
    bound = int(ctx.L.cjs_bz2_compress_bound(total))
    d_out = torch.zeros((bound + 3) & ~3, dtype=torch.uint8, device=dev)
    seg = None
    if world > 1:
        seg = torch.zeros(((bound // world + (1 << 20)) + 3) & ~3, dtype=torch.uint8, device=dev)

    def step():
        if world == 1:
            n = ctx.compress_device(d_in, d_out, args.level)
            return d_out[:n]
        seg.zero_()
        return sharded_compress(ctx, d_in, args.level, seg=seg)

    for _ in range(args.warmup):
        out = step()
    ctx.L.cjs_profile_enable(ctx.h, 1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        out = step()
        dev_ms += ctx.last_device_ms
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    import ctypes as C
    pms, pl, pe = C.c_float(0), C.c_uint32(0), C.c_uint64(0)
    ctx.L.cjs_profile_read(ctx.h, C.byref(pms), C.byref(pl), C.byref(pe))
    ctx.L.cjs_profile_enable(ctx.h, 0)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    if rank == 0:
        comp = out.cpu().numpy().tobytes()
        verified = None
        cpu = None
        if not args.no_verify:
            import bz2
            # independent decoder, bounded to keep the default run short
            limit = min(total, 200_000_000)
            dec = bz2.BZ2Decompressor()
            got = dec.decompress(comp, limit)
            verified = bool(got == host[:limit].tobytes())
            # the whole stream through the GPU decoder (K7-K9), compared on the device
            back = torch.empty(total + 64, dtype=torch.uint8, device=dev)
            nback = ctx.decompress_device(out, back)
            verified = verified and nback == total and bool(torch.equal(back[:total], d_in[:total]))
            del back
            cpu, ref = cpu_baseline(host, args.level, min(args.cpu_sample, total))
            # parity of the leading blocks against the oracle (bit-exact): blocks are encoded
            # independently of what follows, so the oracle's stream of the sample is a prefix of
            # the full stream except for its last block and trailer.
            nfull = (min(args.cpu_sample, total) // (args.level * 100000)) - 1
            if nfull > 0:
                import oracle
                nbytes = 0
                pref = sum(b["bit_len"] for _, b in zip(range(nfull), oracle.block_stages(host[:min(args.cpu_sample, total)], args.level)))
                nbytes = (32 + pref) // 8
                verified = verified and comp[:nbytes] == ref[:nbytes]
        n_launch = max(int(pl.value), 1)
        avg_ms = pms.value / n_launch
        # Dominant kernel: k1_scatter, one stable 8-bit LSD pass over the (key32, index) pair of every
        # rotation of the batch.  ALGORITHMIC bytes per launch = (4+4 read + 4+4 written) = 16 B per
        # block byte; duration = mean of the HIP-event pairs recorded around
        # every launch on the library's own stream during the timed steps.
        alg_bytes = 16.0 * (pe.value / n_launch if pe.value else args.size)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch from the PMC passes of this same command (profiles/r01_pmc_v6_fetch_write.csv:
        # separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs; KiB units; FETCH_SIZE doubled as the
        # MI355X guide prescribes for coalesced streaming reads on gfx950).  Only valid for the
        # default workload/size; null otherwise.
        traffic = None
        if args.workload == "enwik" and args.size == 100_000_000 and world == 1:
            traffic = round((2 * 416.8e6 + 866.2e6), 0)         # profiles/r01_pmc_v6_fetch_write.csv
        line = {
            "metric": "bzip2 -9 compress MB/s on enwik8-shaped input",
            "value": round(total * args.steps / elapsed / 1e6, 2),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic" if args.workload != "e8sa" else "reference test fixtures, tiled",
            "config": {"workload": "synthetic enwik8-shaped text (compressjs_amd.synth.text_like, seed 2025), "
                                   "%d bytes per GPU, bzip2 -%d, %d-byte blocks; BASELINE.json configs[2]"
                                   % (args.size, args.level, args.level * 100000 - 19)
                       if args.workload == "text" else
                       "synthetic enwik8-shaped text with phrase reuse calibrated to enwik8's bzip2 -9 ratio "
                       "(compressjs_amd.synth.enwik_like, seed 2025), %d bytes per GPU, bzip2 -%d; BASELINE.json configs[2]"
                       % (args.size, args.level) if args.workload == "enwik" else
                       "LCG(n, seed 7) random printable ASCII, %d bytes per GPU, bzip2 -%d; BASELINE.json configs[3]"
                       % (args.size, args.level) if args.workload == "lcg" else
                       "test/sample5.ref || test/sample4.ref tiled to %d bytes per GPU (SURVEY.md 8d E8S-A), bzip2 -%d"
                       % (args.size, args.level),
                       "input_bytes": total, "compressed_bytes": len(comp),
                       "blocks_in_flight": args.batch, "sharding": "blocks/%d" % world,
                       "device_ms_per_step": round(dev_ms / args.steps, 3),
                       "bit_exact_vs_oracle_prefix_and_roundtrip": verified,
                       "sha256": hashlib.sha256(comp).hexdigest()},
            "roofline": {"bound": "hbm", "kernel": "k1_scatter", "achieved": round(achieved, 2),
                         "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                         "avg_launch_ms": round(avg_ms, 4), "launches": int(pl.value),
                         "alg_bytes_per_launch": alg_bytes, "traffic": traffic},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
