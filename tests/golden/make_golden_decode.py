#!/usr/bin/env python3
"""Regenerate tests/golden/golden_decode.json: what the REFERENCE decoder (Bzip2.decompressFile /
decompressBlock / table under node 12, /root/reference) does with every stream of
decode_cases.py.  Build-container only.  Usage: python tests/golden/make_golden_decode.py"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import decode_cases  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="golden-dec-")
    jobs, by_id = [], {}
    for k, (sid, s, ms) in enumerate(decode_cases.streams()):
        if s is None:
            raise SystemExit("fixture for %s missing: run in the build container" % sid)
        p = os.path.join(tmp, "%d.bz2" % k)
        open(p, "wb").write(s)
        by_id[sid] = p
        jobs.append(dict(id=sid, kind="unbz2", input=p, multistream=bool(ms)))
    for sid, bitpos in decode_cases.BLOCK_CASES:
        jobs.append(dict(id="block:%s@%d" % (sid, bitpos), kind="unbz2block", input=by_id[sid], bitpos=bitpos))
    jp, rp = os.path.join(tmp, "jobs.json"), os.path.join(tmp, "res.json")
    json.dump(jobs, open(jp, "w"))
    subprocess.check_call(["node", os.path.join(HERE, "ref_runner.js"), jp, rp], timeout=3600)
    res = json.load(open(rp))
    gold = {}
    for r in res:
        gold[r.pop("id")] = r
    meta = dict(made_by="tests/golden/make_golden_decode.py", reference="cscott/compressjs @ /root/reference, node 12",
                n=len(gold))
    json.dump(dict(meta=meta, vectors=gold), open(os.path.join(HERE, "golden_decode.json"), "w"), indent=0, sort_keys=True)
    ok = sum(1 for v in gold.values() if v.get("ok"))
    print("wrote %d vectors (%d decode ok, %d throw)" % (len(gold), ok, len(gold) - ok))


if __name__ == "__main__":
    main()
