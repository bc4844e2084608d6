#!/usr/bin/env python3
"""Pin the MULTI-GPU job streams of bench.py --gpus N (N documents of 10^8 bytes, tests/workloads.py: world_stream) in
tests/golden/golden_big.json, key "<workload>:<N * 10^8>:bz2:9":
  * made_by = "reference": cscott/compressjs under node on the whole job stream (N = 2: ~8 minutes per stream);
  * made_by = "oracle":    oracle/bz2_oracle.c (the C restatement, itself pinned by 159 reference-made vectors and equal to the
                           reference on every 10^8-byte stream) where the reference would take 30-60 minutes per stream (N = 4, 8).
Build container only.  Usage: python tests/golden/make_golden_jobs.py [--workloads enwik lcg e8sa] [--ref-worlds 2] [--oracle-worlds 4 8]"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, "golden_big.json")


def _oracle_job(a):
    w, size, world, level = a
    import workloads
    import oracle
    d = workloads.world_stream(w, size, world)
    t0 = time.time()
    o = oracle.bz2_compress(d, level)
    return ("%s:%d:bz2:%d" % (w, size * world, level),
            dict(kind="bz2", level=level, world=world, made_by="oracle", in_len=int(d.size), in_sha256=hashlib.sha256(d.tobytes()).hexdigest(),
                 out_len=len(o), out_sha256=hashlib.sha256(o).hexdigest(), seconds=round(time.time() - t0, 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="*", default=["enwik", "lcg", "e8sa"])
    ap.add_argument("--size", type=int, default=100_000_000)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--ref-worlds", nargs="*", type=int, default=[2])
    ap.add_argument("--oracle-worlds", nargs="*", type=int, default=[4, 8])
    ap.add_argument("--procs", type=int, default=3)
    args = ap.parse_args()
    import workloads
    res = {}
    tmp = tempfile.mkdtemp(prefix="goldenjobs-")
    refs = []
    for world in args.ref_worlds:
        for w in args.workloads:
            d = workloads.world_stream(w, args.size, world)
            p = os.path.join(tmp, "%s_%d.bin" % (w, world))
            d.tofile(p)
            key = "%s:%d:bz2:%d" % (w, args.size * world, args.level)
            jp, rp = os.path.join(tmp, "j_%s_%d.json" % (w, world)), os.path.join(tmp, "r_%s_%d.json" % (w, world))
            json.dump([dict(id=key, kind="bz2", input=p, level=args.level)], open(jp, "w"))
            refs.append((key, world, subprocess.Popen(["node", "--max-old-space-size=12288", os.path.join(HERE, "ref_runner.js"), jp, rp]), rp))
            print("reference started on", key, flush=True)
            del d
    jobs = [(w, args.size, world, args.level) for world in args.oracle_worlds for w in args.workloads]
    if jobs:
        with mp.get_context("spawn").Pool(args.procs) as pool:
            for key, v in pool.imap_unordered(_oracle_job, jobs):
                res[key] = v
                print("oracle", key, v["out_len"], v["out_sha256"][:12], "%.0f s" % v["seconds"], flush=True)
    for key, world, proc, rp in refs:
        proc.wait()
        if proc.returncode != 0:
            raise SystemExit("reference failed on %s" % key)
        r = json.load(open(rp))[0]
        res[key] = dict(kind="bz2", level=args.level, world=world, made_by="reference", in_len=r["in_len"], in_sha256=r["in_sha256"],
                        out_len=r["out_len"], out_sha256=r["out_sha256"], seconds=round(r["seconds"], 2),
                        mb_per_s=round(r["in_len"] / r["seconds"] / 1e6, 4), nblocks=len(r.get("blocks", [])))
        print("reference", key, r["out_len"], r["out_sha256"][:12], "%.0f s" % r["seconds"], flush=True)
    db = json.load(open(OUT))
    db["vectors"].update(res)
    json.dump(db, open(OUT, "w"), indent=0, sort_keys=True)
    print("wrote", len(res), "vectors")


if __name__ == "__main__":
    main()
