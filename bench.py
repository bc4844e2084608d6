#!/usr/bin/env python3
"""bench.py -- bzip2 -9 compress throughput of the MI355X block pipeline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete bzip2 -9 compression (Bzip2.compressFile equivalent, lib/Bzip2.js:879) of a
10^8-byte stream per GPU (BASELINE.json configs[2]; ~112 blocks of 899 981 bytes), input resident in HBM
when the timed region starts, complete .bz2 stream resident in HBM (rank 0) when it ends.  Workloads
(tests/workloads.py): enwik (default: synthetic enwik8-shaped text with phrase reuse), e8sa (SURVEY.md 8d E8S-A:
the reference's test/sample5.ref || sample4.ref tiled), lcg (configs[3]: random printable ASCII), text, e8sb.
Weak scaling: N GPUs compress an N x 10^8-byte stream; every rank holds its 10^8-byte slice (+ a 3.6 MB margin) only,
every rank plans its own slice (one all_gather of per-slice RLE1 totals, one of the boundary targets the slices hand on: no rank waits for another rank's plan), the encoded segments
are gathered to rank 0 (compressjs_amd/dist.py).  Prints ONE JSON line (rank 0).

What the line carries besides the driver's contract:
  config.bit_exact_vs_reference_digest   sha256 of the WHOLE stream == what the reference itself (node 12) produced
                                         on the same bytes (tests/golden/golden_big.json), all 112 blocks
  config.pcie_inclusive_mb_s             the same step through cjs_bz2_compress (host buffer in, host buffer out), mean of the same number of steps
  config.sample5_mb_s / sample5_ms       BASELINE.json configs[1]: test/sample5.ref (2 MB, three blocks) -9, one host-to-host call
  roofline                               the kernel with the largest total time ON THIS WORKLOAD, found and timed with HIP
                                         events around every launch of K1's main kernels on the library's stream in a
                                         single-stream pass after the timed region (kernel_ms_per_step lists them all);
                                         traffic from the PMC passes of profiles/r06_pmc_traffic.json (stamped with the
                                         build they were collected on; null when none for this workload), e2e = 16 B/B
  cpu_baseline                           kind "reference": Bzip2.compressFile of cscott/compressjs under node, timed ON THIS
                                         BOX IN THIS RUN on the first 10^7 bytes of the same stream (staged copy under
                                         oracle/_ref/refsrc, git-ignored; digest compared with the GPU's for the same prefix);
                                         whole_stream = the same call on all 10^8 bytes, timed in the build container;
                                         all_cores = nproc node processes on equal slices; port = the C restatement
                                         (oracle/) timed live on this box."""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# K1's main kernels (K1P_* classes of csrc/k1_bwt.h) and the ALGORITHMIC HBM bytes one element of a launch costs:
#   k1f_bsort   per rotation: 4 (index read) + 8 (key bytes of the rotation's text) + 4 (suffix-array entry written); since round 5 the kernel
#               reads 16 key bytes per rotation (L2-resident text) and still writes one 8-byte list entry per rotation that ties: the 16 B
#               of rounds 2-4 are kept so that the fractions compare across rounds - a lower bound
#   k1r_round   per list entry and round: 8 (entry read) + 8 (entry written) + 4 (suffix-array entry; the 24 key bytes come from L2-resident text)
#   k1d_build   per rotation: 4 (suffix-array entry read) + 4 (rank written)
#   k1d_round   per list entry and round: 8 (entry read) + 4 (rank gathered) + 8 (entry written)
#   k1d_update  per list entry and round: 8 (entry read) + 4 (rank written) + 8 (entry of the next round's list)
# k1d_med / k1d_large / k1f_task work on groups whose sizes the host never sees: timed, but no byte figure.
KERNELS = [("k1f_bsort", 16.0), ("k1r_round", 20.0), ("k1d_build", 8.0), ("k1d_round", 20.0), ("k1d_med", None), ("k1d_large", None),
           ("k1d_update", 20.0), ("k1f_task", None)]
E2E_ALG_BYTES = 16.0            # 14 + 4 rho + c bytes per input byte, nominal 16 for enwik8-shaped text


def reference_baseline(ctx, host: np.ndarray, level: int, nbytes: int):
    """cscott/compressjs itself (node, the copy __graft_entry__.build() stages under oracle/_ref/refsrc) on the first `nbytes`
    of the stream, one thread, timed by the reference runner (process.hrtime around Bzip2.compressFile) on THIS box in THIS
    run; its output digest is compared with the GPU's for the same bytes.  None when node or the staged copy is missing."""
    import shutil
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "refsrc")
    node = shutil.which("node")
    if not node or not os.path.exists(os.path.join(ref, "main.js")):
        return None
    sample = np.ascontiguousarray(host[:nbytes])
    with tempfile.TemporaryDirectory(prefix="cjsref-") as tmp:
        inp, jobs, res = os.path.join(tmp, "in.bin"), os.path.join(tmp, "jobs.json"), os.path.join(tmp, "res.json")
        sample.tofile(inp)
        json.dump([{"id": "prefix", "kind": "bz2", "input": inp, "level": level}], open(jobs, "w"))
        try:
            subprocess.check_call([node, "--max-old-space-size=4096", os.path.join(ROOT, "tests", "golden", "ref_runner.js"), jobs, res],
                                  env=dict(os.environ, COMPRESSJS_REF=ref), timeout=900, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            r = json.load(open(res))[0]
        except Exception:                                    # noqa: BLE001
            return None
    ours = ctx.compress(sample, level)
    same = bool(hashlib.sha256(ours).hexdigest() == r["out_sha256"] and len(ours) == r["out_len"])
    ver = subprocess.check_output([node, "--version"]).decode().strip()
    return dict(value=round(sample.size / r["seconds"] / 1e6, 4), unit="MB/s", cores=1, kind="reference",
                sample="Bzip2.compressFile(buf, null, %d) of cscott/compressjs under node %s on the first %d bytes of this run's stream, "
                       "%.1f s, timed on this box in this run" % (level, ver, sample.size, r["seconds"]),
                same_bytes_as_gpu=same)


def port_baseline(data: np.ndarray, level: int, sample_bytes: int):
    """The oracle (C restatement of the reference algorithm) timed on ONE host core of THIS box on the first
    `sample_bytes` of the same stream.  Checker code: only this leg may call it."""
    import oracle
    sample = data[:sample_bytes]
    t0 = time.perf_counter()
    out = oracle.bz2_compress(sample, level)
    dt = time.perf_counter() - t0
    return dict(value=round(sample.size / dt / 1e6, 4), unit="MB/s", cores=1,
                sample="first %d bytes of the same stream, oracle/bz2_oracle.c, this box" % sample.size), out


def bench_bwtc(args):
    """Secondary line (not the driver's default): BWTC -9 on one GPU, host buffers in and out."""
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1, "BWTC does not shard (serial range coder): N = 1"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path)"
    import workloads
    from compressjs_amd.bzip2 import Context
    wl = args.workload if args.workload != "enwik" else "e8sa"           # BWTC -9 on E8S-A
    host = workloads.stream(wl, args.size)
    ctx = Context(0, args.batch)
    for _ in range(max(args.warmup, 1)):
        out = ctx.bwtc_compress(host, args.level)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = ctx.bwtc_compress(host, args.level)
    dt = (time.perf_counter() - t0) / args.steps
    sha = hashlib.sha256(out).hexdigest()
    gold = {}
    gpath = os.path.join(ROOT, "tests", "golden", "golden_big.json")
    if os.path.exists(gpath):
        gold = json.load(open(gpath))["vectors"]
    g = gold.get("%s:%d:bwtc:%d" % (wl, args.size, args.level))
    vs_ref = None if g is None else bool(g["out_sha256"] == sha and g["out_len"] == len(out))
    # the same stream with the model on the host (round-1 path) must give the same bytes; and back through BWTC.decompressFile
    back = None
    if not args.no_verify and args.size <= 30_000_000:
        back = bool(ctx.bwtc_decompress(np.frombuffer(out, dtype=np.uint8)) == host.tobytes())
    # the dominant GPU kernel of this path is k10_model (the adaptive FenwickModel: one wave per block, a serial recurrence): its
    # algorithmic bytes are 2 (symbol read) + 8 (triple written) per encodeFreq call - measured live with HIP events around the launch
    import ctypes as C
    tms = (C.c_float * 5)()
    ctx.L.cjs_bwtc_last_times(ctx.h, tms)
    k10_ms, landed_ms, coder_ms, total_ms, ncalls = [float(x) for x in tms]
    roof = None
    if k10_ms > 0:
        ach = 10.0 * ncalls / (k10_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k10_model", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 6),
                "avg_launch_ms": round(k10_ms, 2), "alg_bytes_per_launch": 10.0 * ncalls, "traffic": None,
                "note": "not bandwidth-bound: one wave per block walks a serial recurrence, ~%d clocks per symbol (tests/microbench/lone_wave.hip: "
                        "5-9 clocks per dependent instruction, ~50 per LDS round trip for a wave alone on its SIMD); the line's value is bound by the "
                        "serial host range coder: %.1f ns per encodeFreq call over %.1f M calls = %.0f ms of the %.0f ms step, after %.0f ms of GPU stages"
                        % (round(k10_ms * 1e-3 * 2.4e9 / max(ncalls / max(1, (args.size + args.level * 100000 - 1) // (args.level * 100000)), 1)),
                           coder_ms * 1e6 / max(ncalls, 1), ncalls / 1e6, coder_ms, total_ms, landed_ms)}
    print(json.dumps({
        "metric": "BWTC -9 compress MB/s (BASELINE.json configs[4])", "value": round(args.size / dt / 1e6, 2), "unit": "MB/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic" if wl not in ("e8sa", "e8sb") else "reference test fixtures, tiled",
        "config": {"workload": "%s, %d bytes, BWTC -%d, %d-byte blocks, host buffers in and out" % (workloads.DESCRIPTIONS[wl], args.size, args.level, args.level * 100000),
                   "compressed_bytes": len(out), "sha256": sha, "bit_exact_vs_reference_digest": vs_ref, "roundtrip": back,
                   "gpu_stages": "BWT.bwtransform (K1 linear), MTF/RLE2 (K2), FenwickModel (K10); host: RangeCoder.encodeFreq, serial"},
        "roofline": roof,
        "cpu_baseline": None if g is None else {"value": g["mb_per_s"], "unit": "MB/s", "cores": 1, "kind": "reference",
                                                "sample": "BWTC.compressFile(buf, null, %d) of cscott/compressjs under node 12 on the same %d bytes, %.1f s, build container" % (args.level, g["in_len"], g["seconds"])}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)       # (a 13 ms step: the clocks of an idle GPU need several to come up)
    ap.add_argument("--size", type=int, default=100_000_000, help="input bytes per GPU")
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--workload", default="enwik", choices=["enwik", "text", "lcg", "e8sa", "e8sb"])
    ap.add_argument("--cpu-sample", type=int, default=12_000_000)
    ap.add_argument("--ref-sample", type=int, default=10_000_000, help="bytes of the stream the reference (node) is timed on, on this box")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--full-verify", action="store_true", help="N > 1: rebuild the whole job stream on rank 0 and decode / compare it even when a pinned digest exists")
    ap.add_argument("--batch", type=int, default=128, help="bzip2 blocks in flight (over all streams)")
    ap.add_argument("--force-fallback", action="store_true", help="N > 1, test rigs: take the replicated fall-back plan on an input that plans slice by slice")
    ap.add_argument("--codec", default="bz2", choices=["bz2", "bwtc"],
                    help="bwtc: BWTC.compressFile -9 (BASELINE.json configs[4]; N = 1, host buffers in and out: its range coder is serial "
                         "host code, lib/RangeCoder.js; BWT, MTF/RLE2 and the adaptive FenwickModel run on the GPU)")
    args = ap.parse_args()
    if args.codec == "bwtc":
        return bench_bwtc(args)

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

    import workloads
    from compressjs_amd.bzip2 import Context
    from compressjs_amd.dist import margin_bytes, sharded_compress, sharded_compress_parallel, slice_bounds

    total = args.size * world
    ctx = Context(local, args.batch)
    on_device = args.workload == "lcg" and world > 1           # cfg4: every GPU fills its own slice in HBM (LCG jump-ahead)
    host = None
    if world == 1:
        host = workloads.stream(args.workload, total)
        d_in = torch.from_numpy(host).to(dev)
    else:
        # the job's stream = one document of args.size bytes per rank (tests/workloads.py): a rank generates its own document and
        # the margin in front of it, not the whole job; rank 0 names the whole stream only after the timed region, to verify it
        lo, hi = slice_bounds(total, rank, world)
        whi = min(total, hi + margin_bytes(args.level))        # the slice + the head of what follows it (parallel plan)
        if on_device:
            d_in = torch.empty(whi - lo, dtype=torch.uint8, device=dev)
            ctx.lcg_ascii_device(d_in, 7, first=lo)
        else:
            win = workloads.window_after(args.workload, args.size, rank, margin_bytes(args.level), world)
            assert win.size == whi - lo
            d_in = torch.from_numpy(win).to(dev)
            del win
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
        def whole():
            # only the replicated fall-back needs the whole stream on every rank (a block boundary inside a run of four or more equal bytes
            # somewhere in the job - 0.05 % per boundary on the enwik-shaped streams): the ranks' own slices, all_gathered on the devices
            # (round 5; until then every rank named the whole job on its host again, minutes inside the timed region)
            if on_device:
                t = torch.empty(total, dtype=torch.uint8, device=dev)
                ctx.lcg_ascii_device(t, 7, first=0)
                return t
            own = d_in[:hi - lo].contiguous()
            if backend != "nccl":
                own = own.cpu()                                # (gloo test rigs: collectives on host tensors)
            parts = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(parts, own)
            return torch.cat(parts).to(dev)
        return sharded_compress_parallel(ctx, d_in, hi - lo, lo, total, args.level, seg=seg,
                                         fallback=lambda: sharded_compress(ctx, whole(), args.level), force_fallback=args.force_fallback)

    for _ in range(args.warmup):
        out = step()
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

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    if rank == 0:
        import ctypes as C
        comp = out.cpu().numpy().tobytes()
        sha = hashlib.sha256(comp).hexdigest()
        # ---- parity of the WHOLE stream: the digest the reference itself produced on these bytes -------------
        gold = {}
        gpath = os.path.join(ROOT, "tests", "golden", "golden_big.json")
        if os.path.exists(gpath):
            gold = json.load(open(gpath))["vectors"]
        gkey = "%s:%d:bz2:%d" % (args.workload, total, args.level)
        g = gold.get(gkey)
        vs_ref = None
        # N > 1 with a pinned digest (tests/golden/make_golden_jobs.py: reference-made at N = 2, 4 and 8 since round 6 - the line says which in reference_digest_made_by): the digest
        # IS the check; naming the whole N-document stream on rank 0's host again takes minutes and is only done without one
        digest_only = world > 1 and g is not None and not args.full_verify
        if host is None and not digest_only:
            host = workloads.world_stream(args.workload, args.size, world)
        if g is not None:
            if host is not None:
                assert hashlib.sha256(host.tobytes()).hexdigest() == g["in_sha256"], "workload generator drifted from the reference-made golden"
            vs_ref = bool(sha == g["out_sha256"] and len(comp) == g["out_len"])
        verified, port, pcie, decode_mb_s = None, None, None, None
        prefix_ok = None
        if digest_only and not args.no_verify:
            # N > 1 with a pinned digest: besides the digest, an independent decoder (libbz2) on a bounded prefix of the assembled stream
            # against what rank 0 itself holds - its own document, the first args.size bytes of the job
            import bz2
            limit = min(args.size, 50_000_000)
            mine = d_in[:limit].cpu().numpy().tobytes()
            try:
                prefix_ok = bool(bz2.BZ2Decompressor().decompress(comp, limit) == mine)
            except Exception:                                # noqa: BLE001
                prefix_ok = False
        if not args.no_verify and not digest_only:
            import bz2
            # independent decoder (libbz2), bounded to keep the default run short
            limit = min(total, 200_000_000)
            verified = bool(bz2.BZ2Decompressor().decompress(comp, limit) == host[:limit].tobytes())
            # the whole stream through the GPU decoder (K7-K9), compared on the device
            back = torch.empty(total + 64, dtype=torch.uint8, device=dev)
            nback = ctx.decompress_device(out, back)
            torch.cuda.synchronize()
            a = time.perf_counter()                            # (the second call: the decoder's buffers exist now)
            nback = ctx.decompress_device(out, back)
            torch.cuda.synchronize()
            decode_mb_s = round(total / (time.perf_counter() - a) / 1e6, 1)
            whole = d_in if world == 1 else torch.from_numpy(host).to(dev)
            verified = verified and nback == total and bool(torch.equal(back[:total], whole[:total]))
            del back, whole
            # localiser: the leading blocks bit for bit against the oracle (says WHICH block differs if the digest does)
            port, ref = port_baseline(host, args.level, min(args.cpu_sample, total))
            nfull = (min(args.cpu_sample, total) // (args.level * 100000)) - 1
            if nfull > 0:
                import oracle
                pref = sum(b["bit_len"] for _, b in zip(range(nfull), oracle.block_stages(host[:min(args.cpu_sample, total)], args.level)))
                nbytes = (32 + pref) // 8
                verified = verified and comp[:nbytes] == ref[:nbytes]
            if world == 1:
                # host buffer in -> .bz2 in host memory through the C ABI (SURVEY.md 8d's end-to-end definition): the MEAN of the same
                # number of steps as the headline (round 6; until then the minimum of three calls - not the estimator `value` is)
                hbuf = np.zeros(bound, dtype=np.uint8)             # caller-owned output buffer, pages touched
                nn = 0
                for _ in range(max(1, args.warmup // 2)):
                    nn = int(ctx.L.cjs_bz2_compress(ctx.h, host.ctypes.data, host.size, args.level, hbuf.ctypes.data, hbuf.size))
                a = time.perf_counter()
                for _ in range(args.steps):
                    nn = int(ctx.L.cjs_bz2_compress(ctx.h, host.ctypes.data, host.size, args.level, hbuf.ctypes.data, hbuf.size))
                pcie_dt = (time.perf_counter() - a) / args.steps
                assert nn == len(comp) and hashlib.sha256(hbuf[:nn].tobytes()).hexdigest() == sha
                pcie = round(total / pcie_dt / 1e6, 1)
        # ---- roofline leg: K1's main kernels alone on the GPU (one stream), HIP events around every launch; the one with the
        #      largest total on this workload is the line's dominant kernel -----------------------------------------------
        prof_steps, kms, dom = 3, {}, None
        avg_ms, launches, elements, alg_per_el = 0.0, 0, 0, None
        if world == 1:
            prev_streams = os.environ.get("CJS_STREAMS")           # (restored below: the E8S-A and PCIe legs run under the caller's setting)
            os.environ["CJS_STREAMS"] = "1"
            ctx1 = Context(local, args.batch)
            ctx1.compress_device(d_in, d_out, args.level)
            ctx1.L.cjs_profile_enable(ctx1.h, 1)
            for _ in range(prof_steps):
                ctx1.compress_device(d_in, d_out, args.level)
            torch.cuda.synchronize()
            rows = []
            for cls, (name, per_el) in enumerate(KERNELS):
                pms, pl, pe = C.c_float(0), C.c_uint32(0), C.c_uint64(0)
                ctx1.L.cjs_profile_read_class(ctx1.h, cls, C.byref(pms), C.byref(pl), C.byref(pe))
                rows.append((name, per_el, float(pms.value), int(pl.value), int(pe.value)))
                kms[name] = round(pms.value / prof_steps, 4)
            ctx1.L.cjs_profile_enable(ctx1.h, 0)
            ctx1.close()
            if prev_streams is None:
                os.environ.pop("CJS_STREAMS", None)
            else:
                os.environ["CJS_STREAMS"] = prev_streams
            dom, alg_per_el, tot_ms, launches, elements = max(rows, key=lambda r: r[2])
            avg_ms = tot_ms / max(launches, 1)
        alg_bytes = None if (alg_per_el is None or not launches) else alg_per_el * elements / launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if (alg_bytes and avg_ms > 0) else None
        # HBM traffic per launch: FETCH_SIZE / WRITE_SIZE passes of this workload (separate --pmc runs, tests/gpu_r6_traffic.sh;
        # FETCH_SIZE doubled as the MI355X guide prescribes for gfx950), committed as profiles/r06_pmc_traffic.json together with
        # the build they were collected on.  null when no pass for this workload / size / kernel is committed.
        traffic, traffic_build = None, None
        tpath = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
        if os.path.exists(tpath) and dom:
            tj = json.load(open(tpath))
            ent = tj.get("%s:%d" % (args.workload, args.size), {}).get(dom)
            if ent and launches:
                traffic = round(ent["traffic_bytes_per_step"] * prof_steps / launches)
                traffic_build = tj.get("build")
        # ---- SURVEY.md 8(d)'s real-text headline (E8S-A: test/sample5.ref || sample4.ref tiled) in the same line --------------
        e8 = {"mb_s": None, "ms": None, "exact": None}
        if world == 1 and args.workload != "e8sa" and workloads.have_fixtures() and not args.no_verify:
            h2 = workloads.stream("e8sa", args.size)
            d2 = torch.from_numpy(h2).to(dev)
            for _ in range(args.warmup):                       # (the headline's own warm-up and step counts)
                n2 = ctx.compress_device(d2, d_out, args.level)
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(args.steps):
                n2 = ctx.compress_device(d2, d_out, args.level)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - a) / args.steps
            o2 = d_out[:n2].cpu().numpy().tobytes()
            g2 = gold.get("e8sa:%d:bz2:%d" % (args.size, args.level))
            e8 = {"mb_s": round(args.size / dt2 / 1e6, 2), "ms": round(dt2 * 1e3, 3),
                  "exact": None if g2 is None else bool(hashlib.sha256(o2).hexdigest() == g2["out_sha256"] and len(o2) == g2["out_len"])}
            del d2, h2
        # ---- BASELINE.json configs[1]: test/sample5.ref (2 130 640 bytes, three blocks), bzip2 -9, ONE call through cjs_bz2_compress
        #      (host buffer in, host buffer out): the small-input rate, bound by the latency of the kernel chain, not by throughput
        s5 = {"mb_s": None, "ms": None, "exact": None}
        s5path = os.path.join(ROOT, "oracle", "_ref", "fixtures", "sample5.ref")
        if world == 1 and os.path.exists(s5path) and not args.no_verify:
            h5 = np.fromfile(s5path, dtype=np.uint8)
            b5 = np.zeros(int(ctx.L.cjs_bz2_compress_bound(h5.size)), dtype=np.uint8)
            n5 = 0
            for _ in range(3):
                n5 = int(ctx.L.cjs_bz2_compress(ctx.h, h5.ctypes.data, h5.size, args.level, b5.ctypes.data, b5.size))
            a = time.perf_counter()
            for _ in range(args.steps):
                n5 = int(ctx.L.cjs_bz2_compress(ctx.h, h5.ctypes.data, h5.size, args.level, b5.ctypes.data, b5.size))
            dt5 = (time.perf_counter() - a) / args.steps
            # the reference's own output for sample5.ref at -9 is 274 768 bytes, sha256 236be53b...
            s5 = {"mb_s": round(h5.size / dt5 / 1e6, 2), "ms": round(dt5 * 1e3, 3),
                  "exact": bool(args.level == 9 and n5 == 274768 and hashlib.sha256(b5[:n5].tobytes()).hexdigest().startswith("236be53b"))}
        wall = elapsed / args.steps
        ref_line = None
        if world == 1 and not args.no_verify:
            ref_line = reference_baseline(ctx, host, args.level, min(args.ref_sample, total))
        whole = None
        if g is not None and "mb_per_s" in g and world == 1:
            ac = gold.get("%s:%d:allcores:%d" % (args.workload, total, args.level))       # this workload's own all-cores timing, or none
            whole = {"value": g["mb_per_s"], "unit": "MB/s", "cores": 1,
                     "sample": "the same call on the whole %d-byte stream of this run (same sha256 in and out as this line), %.1f s, "
                               "timed in the build container (tests/golden/make_golden_big.py)" % (g["in_len"], g["seconds"]),
                     "all_cores": None if not ac else {"value": ac["sum_mb_per_s"], "unit": "MB/s", "processes": ac["processes"],
                                                       "sample": "%d node processes on equal slices of this stream, summed" % ac["processes"]}}
        if ref_line is not None:
            ref_line["whole_stream"] = whole
            ref_line["port"] = port
        elif whole is not None:
            ref_line = dict(whole, kind="reference", port=port)
            ref_line["sample"] = ("Bzip2.compressFile(buf, null, %d) of cscott/compressjs under node 12: " % args.level) + ref_line["sample"] + \
                                 "; no node / staged reference on this box"
        elif port is not None:
            ref_line = dict(port, kind="port")
        line = {
            "metric": "bzip2 -9 compress MB/s on enwik8-shaped input",
            "value": round(total * args.steps / elapsed / 1e6, 2),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if args.workload not in ("e8sa", "e8sb") else "reference test fixtures, " + ("tiled" if args.workload == "e8sa" else "order-2 Markov"),
            "config": {"workload": "%s, %d bytes per GPU, bzip2 -%d, %d-byte blocks; BASELINE.json configs[%d]"
                                   % (workloads.DESCRIPTIONS[args.workload], args.size, args.level, args.level * 100000 - 19,
                                      3 if args.workload == "lcg" else 2),
                       "input_bytes": total, "compressed_bytes": len(comp),
                       "blocks_in_flight": args.batch,
                       "sharding": "blocks/%d" % world if world == 1 else
                                   "one %d-byte document per GPU, one .bz2 stream of the %d documents; every rank holds its document + %d bytes of the next one (parallel plan: one all_gather of per-slice RLE1 totals, one of the slices' boundary targets; no rank waits for another rank's plan)"
                                   % (args.size, world, margin_bytes(args.level)),
                       "device_ms_per_step": round(dev_ms / args.steps, 3),
                       "bit_exact_vs_reference_digest": vs_ref,
                       "reference_digest_made_by": None if g is None else g.get("made_by", "reference"),
                       "e8sa_mb_s": e8["mb_s"], "e8sa_ms_per_step": e8["ms"], "e8sa_bit_exact_vs_reference_digest": e8["exact"],
                       "bit_exact_vs_oracle_prefix_and_roundtrip": verified,
                       "libbz2_prefix_roundtrip": prefix_ok,
                       "pcie_inclusive_mb_s": pcie,
                       "sample5_mb_s": s5["mb_s"], "sample5_ms": s5["ms"], "sample5_bit_exact_vs_reference_digest": s5["exact"],
                       "gpu_decode_mb_s": decode_mb_s,
                       "sha256": sha},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": None if achieved is None else round(achieved, 2),
                         "peak": 8000.0, "unit": "GB/s", "frac": None if achieved is None else round(achieved / 8000.0, 5),
                         "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                         "alg_bytes_per_launch": alg_bytes, "traffic": traffic, "traffic_build": traffic_build,
                         "kernel_ms_per_step": kms,
                         "e2e": {"alg_bytes_per_input_byte": E2E_ALG_BYTES,
                                 "achieved": round(E2E_ALG_BYTES * total / wall / 1e9, 2),
                                 "frac": round(E2E_ALG_BYTES * total / wall / 8e12, 5)}},
            "cpu_baseline": ref_line,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
