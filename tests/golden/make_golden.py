#!/usr/bin/env python3
"""Regenerate tests/golden/golden.json by running the REFERENCE (cscott/compressjs under node,
/root/reference) on the inputs of cases.py.  Build-container only; the GPU box and the tests
never execute this.  Usage: python tests/golden/make_golden.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import cases  # noqa: E402


def huff_cases():
    """Sorted frequency vectors for allocateHuffmanCodeLengths (HuffmanAllocator.js:199):
    the reference's own KATs (test/huffman.js:15-77) plus differential-fuzz vectors that
    reach the relocation branch (zero-heavy tails, Fibonacci-like skews)."""
    fib = [0, 1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181,
           6765, 10946, 17711, 28657, 46368, 75025, 121393, 196418, 317811, 514229, 832040,
           1346269, 2178309, 3524578, 5702887, 9227465, 14930352]
    out = [dict(freq=[1], max_len=32), dict(freq=[1, 1], max_len=32),
           dict(freq=[1] * 5, max_len=32), dict(freq=[0, 0, 1, 1, 1, 1], max_len=3),
           dict(freq=fib[:36], max_len=20), dict(freq=fib[:22], max_len=20),
           dict(freq=fib[:21], max_len=20), dict(freq=fib[:36], max_len=6)]
    rng = np.random.RandomState(12345)
    for t in range(400):
        n = int(rng.choice([3, 4, 5, 7, 16, 33, 64, 100, 150, 200, 257, 258]))
        style = t % 5
        if style == 0:
            f = rng.randint(0, 1000, size=n)
        elif style == 1:   # many zeros under a skewed tail (SURVEY.md section 7 hard part 4)
            f = np.concatenate([np.zeros(n // 2, dtype=np.int64),
                                (1.5 ** np.arange(n - n // 2)).astype(np.int64) % 900000])
        elif style == 2:   # geometric
            f = (rng.uniform(1.2, 2.2) ** (np.arange(n) * 24.0 / n)).astype(np.int64)
        elif style == 3:   # zipf
            f = (900000.0 / (1 + np.arange(n)) ** rng.uniform(0.8, 2.5)).astype(np.int64)
        else:              # all equal / ones with a few spikes
            f = np.ones(n, dtype=np.int64) * int(rng.randint(0, 3))
            f[rng.randint(0, n, size=3)] = rng.randint(0, 900000, size=3)
        f = np.sort(f)[:n]
        out.append(dict(freq=[int(x) for x in f], max_len=20))
    return out


def main():
    tmp = tempfile.mkdtemp(prefix="golden-")
    jobs = []
    for cid, (_, levels) in cases.CASES.items():
        data = cases.case_input(cid)
        if data is None:
            raise SystemExit("fixture for %s missing: run in the build container" % cid)
        p = os.path.join(tmp, cid + ".bin")
        data.tofile(p)
        for lv in levels:
            jobs.append(dict(id="%s:bz2:%d" % (cid, lv), kind="bz2", input=p, level=lv))
        jobs.append(dict(id="%s:crc" % cid, kind="crc", input=p))
        if cid in cases.BWT_CASES:
            jobs.append(dict(id="%s:bwt2" % cid, kind="bwt2", input=p))
            jobs.append(dict(id="%s:bwt" % cid, kind="bwt", input=p))
            jobs.append(dict(id="%s:sa" % cid, kind="sa", input=p))
        for pidx in cases.UNBWT_CASES.get(cid, []):
            jobs.append(dict(id="%s:unbwt:%d" % (cid, pidx), kind="unbwt", input=p, pidx=pidx))
        if cid in ("sample0", "sample1", "sample3", "empty", "a1000", "text100k"):
            jobs.append(dict(id="%s:bwtc:9" % cid, kind="bwtc", input=p, level=9))
        for bc, lv in (("text950k", 9), ("text100k", 7), ("bytes40", 6), ("runs300k", 8), ("lcg250000", 9),
                       ("a1", 9), ("text1k", 9), ("sample2", 6), ("text2500k", 8), ("zeros300k", 9),
                       # levels 1-5: DefSumModel instead of FenwickModel (lib/BWTC.js:107)
                       ("text100k", 3), ("bytes40", 1), ("runs300k", 5), ("a1000", 2), ("lcg250000", 4), ("empty", 1),
                       ("text950k", 5), ("sample1", 2), ("zeros300k", 1), ("text1k", 4)):
            if cid == bc:
                jobs.append(dict(id="%s:bwtc:%d" % (cid, lv), kind="bwtc", input=p, level=lv))
    jobs.append(dict(id="huff", kind="huff", cases=huff_cases()))
    jp, rp = os.path.join(tmp, "jobs.json"), os.path.join(tmp, "res.json")
    json.dump(jobs, open(jp, "w"))
    subprocess.check_call(["node", os.path.join(HERE, "ref_runner.js"), jp, rp])
    res = json.load(open(rp))
    gold = {r["id"]: r for r in res}
    for r in gold.values():
        r.pop("id")
        if "seconds" in r:
            r["seconds"] = round(r["seconds"], 3)
    meta = dict(node=subprocess.check_output(["node", "--version"]).decode().strip(),
                reference="cscott/compressjs @ /root/reference (package.json version 1.0.3-git)")
    json.dump(dict(meta=meta, vectors=gold), open(os.path.join(HERE, "golden.json"), "w"),
              indent=0, sort_keys=True)
    print("wrote", len(gold), "vectors")


if __name__ == "__main__":
    main()
