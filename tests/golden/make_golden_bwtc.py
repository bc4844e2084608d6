#!/usr/bin/env python3
"""tests/golden/golden_bwtc.json: the REFERENCE's BWTC output (length + sha256) for the fuzz inputs of bwtc_cases.py and for
SURVEY.md 8(c)'s BWTC -9 digests of test/sample2/4/5.ref.  Build container only (node + /root/reference).
Usage: python tests/golden/make_golden_bwtc.py"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import bwtc_cases  # noqa: E402
import cases  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="golden-bwtc-")
    jobs = []
    for i in range(bwtc_cases.N_SMALL):
        d, lv = bwtc_cases.case(i)
        p = os.path.join(tmp, "c%d.bin" % i)
        d.tofile(p)
        jobs.append(dict(id="fuzz%d" % i, kind="bwtc", input=p, level=lv))
    for cid, d, lv in bwtc_cases.big_cases():
        p = os.path.join(tmp, cid + ".bin")
        d.tofile(p)
        jobs.append(dict(id=cid, kind="bwtc", input=p, level=lv))
    for cid in ("sample2", "sample4", "sample5"):
        d = cases.case_input(cid)
        if d is None:
            raise SystemExit("fixture for %s missing: run in the build container" % cid)
        p = os.path.join(tmp, cid + ".bin")
        d.tofile(p)
        jobs.append(dict(id="%s:bwtc:9" % cid, kind="bwtc", input=p, level=9))
    jp, rp = os.path.join(tmp, "jobs.json"), os.path.join(tmp, "res.json")
    json.dump(jobs, open(jp, "w"))
    subprocess.check_call(["node", os.path.join(HERE, "ref_runner.js"), jp, rp])
    gold = {}
    for r in json.load(open(rp)):
        gold[r["id"]] = dict(level=r["level"], in_len=r["in_len"], in_sha256=r["in_sha256"], out_len=r["out_len"], out_sha256=r["out_sha256"])
    meta = dict(node=subprocess.check_output(["node", "--version"]).decode().strip(),
                reference="cscott/compressjs @ /root/reference (package.json version 1.0.3-git)")
    json.dump(dict(meta=meta, vectors=gold), open(os.path.join(HERE, "golden_bwtc.json"), "w"), indent=0, sort_keys=True)
    print("wrote", len(gold), "vectors")


if __name__ == "__main__":
    main()
