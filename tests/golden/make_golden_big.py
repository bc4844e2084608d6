#!/usr/bin/env python3
"""Pin the FULL-SIZE bench streams to the reference: runs cscott/compressjs (node, /root/reference)
on the 10^8-byte streams of tests/workloads.py at bzip2 -9 and records digest, length and the
reference's own wall time (process.hrtime around Bzip2.compressFile) in tests/golden/golden_big.json.
Also times the "all host cores" row of SURVEY.md 8(d): nproc node processes on equal slices.

Build-container only (about 5 minutes of one core per stream); the GPU box and the tests never run
this, they read golden_big.json.  Usage:
    python tests/golden/make_golden_big.py [workload ...] [--size N] [--allcores WORKLOAD]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import workloads  # noqa: E402

OUT = os.path.join(HERE, "golden_big.json")


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_ref(jobs, tmp, tag):
    jp, rp = os.path.join(tmp, "jobs_%s.json" % tag), os.path.join(tmp, "res_%s.json" % tag)
    json.dump(jobs, open(jp, "w"))
    return subprocess.Popen(["node", "--max-old-space-size=8192", os.path.join(HERE, "ref_runner.js"), jp, rp]), rp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["enwik", "e8sa", "lcg"])
    ap.add_argument("--size", type=int, default=100_000_000)
    ap.add_argument("--level", type=int, default=9)
    ap.add_argument("--allcores", default="enwik", help="workload for the nproc-processes row ('' = skip)")
    ap.add_argument("--parallel", type=int, default=2, help="reference processes at a time (full streams)")
    ap.add_argument("--codec", default="bz2", choices=["bz2", "bwtc"], help="bwtc: BWTC.compressFile (BASELINE.json configs[4])")
    args = ap.parse_args()
    db = json.load(open(OUT)) if os.path.exists(OUT) else dict(meta={}, vectors={})
    db["meta"] = dict(node=subprocess.check_output(["node", "--version"]).decode().strip(),
                      reference="cscott/compressjs @ /root/reference (package.json version 1.0.3-git)",
                      cpu=cpu_model(), nproc=os.cpu_count(),
                      note="seconds = process.hrtime around Bzip2.compressFile(buf, null, level) in the build container, one thread")
    tmp = tempfile.mkdtemp(prefix="goldenbig-")
    pending = []
    for w in args.workloads:
        data = workloads.stream(w, args.size)
        p = os.path.join(tmp, "%s.bin" % w)
        data.tofile(p)
        key = "%s:%d:%s:%d" % (w, args.size, args.codec, args.level)
        pending.append((key, [dict(id=key, kind=args.codec, input=p, level=args.level)]))
        del data
    running = []
    while pending or running:
        while pending and len(running) < args.parallel:
            key, jobs = pending.pop(0)
            proc, rp = run_ref(jobs, tmp, key.replace(":", "_"))
            running.append((key, proc, rp))
            print("started", key, flush=True)
        for it in list(running):
            key, proc, rp = it
            if proc.poll() is None:
                continue
            running.remove(it)
            if proc.returncode != 0:
                raise SystemExit("reference failed on %s" % key)
            r = json.load(open(rp))[0]
            r.pop("id")
            if "blocks" in r:
                r["nblocks"] = len(r.pop("blocks"))
            r["seconds"] = round(r["seconds"], 2)
            r["mb_per_s"] = round(r["in_len"] / r["seconds"] / 1e6, 4)
            db["vectors"][key] = r
            json.dump(db, open(OUT, "w"), indent=0, sort_keys=True)
            print("done", key, r["out_len"], r["out_sha256"][:16], r["seconds"], "s", flush=True)
        time.sleep(2)
    if args.allcores:
        # "fork nproc Node processes on equal slices and sum MB/s"
        w, nproc = args.allcores, os.cpu_count()
        data = workloads.stream(w, args.size)
        per = args.size // nproc
        procs = []
        t0 = time.time()
        for i in range(nproc):
            p = os.path.join(tmp, "%s_slice%d.bin" % (w, i))
            data[i * per:(i + 1) * per].tofile(p)
            procs.append(run_ref([dict(id="s%d" % i, kind="bz2", input=p, level=args.level)], tmp, "slice%d" % i))
        rates, secs = [], []
        for proc, rp in procs:
            if proc.wait() != 0:
                raise SystemExit("reference failed on a slice")
            r = json.load(open(rp))[0]
            secs.append(r["seconds"])
            rates.append(r["in_len"] / r["seconds"] / 1e6)
        db["vectors"]["%s:%d:allcores:%d" % (w, args.size, args.level)] = dict(
            processes=nproc, slice_bytes=per, seconds_each=[round(s, 2) for s in secs],
            sum_mb_per_s=round(sum(rates), 4), in_sha256=hashlib.sha256(data.tobytes()).hexdigest())
        json.dump(db, open(OUT, "w"), indent=0, sort_keys=True)
        print("allcores", nproc, "processes:", round(sum(rates), 3), "MB/s", flush=True)


if __name__ == "__main__":
    main()
