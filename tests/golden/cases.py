"""Catalogue of parity inputs.  Every input is either synthesised deterministically
(compressjs_amd.synth) or one of the reference's own test fixtures (test/sample*.ref), which
are looked up under /root/reference/test (build container) or oracle/_ref/fixtures (staged
there by __graft_entry__.build(); git-ignored, travels with gpurun).  Fixtures are never
committed."""
from __future__ import annotations

import os
import numpy as np

from compressjs_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIXTURE_DIRS = ["/root/reference/test", os.path.join(ROOT, "oracle", "_ref", "fixtures")]


def _lcg_plus(n, seed, tail):
    return np.concatenate([synth.lcg_ascii(n, seed), np.frombuffer(tail, dtype=np.uint8)])


# id -> (builder, bz2 levels to pin)
CASES = {
    # reference fixtures (SURVEY.md 8c digests)
    "sample0": (("fixture", "sample0.ref"), [1, 9]),
    "sample1": (("fixture", "sample1.ref"), [1, 9]),
    "sample2": (("fixture", "sample2.ref"), [1, 9]),
    "sample3": (("fixture", "sample3.ref"), [1, 9]),
    "sample4": (("fixture", "sample4.ref"), [1, 9]),
    "sample5": (("fixture", "sample5.ref"), [1, 9]),
    # crafted edge cases
    "empty": (("bytes", b""), [1, 9]),
    "a1": (("bytes", b"a"), [9]),
    "a3": (("bytes", b"a" * 3), [9]),
    "a4": (("bytes", b"a" * 4), [9]),
    "a5": (("bytes", b"a" * 5), [9]),
    "a8": (("bytes", b"a" * 8), [9]),
    "a255": (("bytes", b"a" * 255), [9]),
    "a256": (("bytes", b"a" * 256), [9]),
    "a259": (("bytes", b"a" * 259), [9]),
    "a260": (("bytes", b"a" * 260), [9]),
    "a1000": (("bytes", b"a" * 1000), [9]),
    "ab500": (("bytes", b"ab" * 500), [9]),
    "abc_tie": (("bytes", b"abcabc"), [9]),
    "banana": (("bytes", b"banana"), [9]),
    "bytes40": (("all_bytes", 40), [9]),
    "mary9": (("bytes", b"Mary had a little lamb, its fleece was white as snow" * 8
               + b"Nary had a little lamb, its fleece was white as snow"), [9]),
    # block-boundary cases: RLE1 state at the moment the block fills (Bzip2.js:636-667)
    "lcg99977_a10": (("lcg_plus", 99977, 1, b"a" * 10), [1]),
    "lcg99978_a10": (("lcg_plus", 99978, 1, b"a" * 10), [1]),
    "lcg99976_a300": (("lcg_plus", 99976, 1, b"a" * 300), [1]),
    "lcg99981": (("lcg", 99981, 1), [1]),
    "lcg99982": (("lcg", 99982, 1), [1]),
    "lcg250000": (("lcg", 250000, 7), [1]),
    "lcg2000000": (("lcg", 2000000, 7), [9]),
    # synthetic streams
    "text1k": (("text", 1000, 1), [9]),
    "text100k": (("text", 100000, 2), [1, 9]),
    "text950k": (("text", 950000, 3), [9]),
    "text2500k": (("text", 2500000, 4), [9, 3]),
    "runs300k": (("runs", 300000, 3), [1, 9]),
    "runs1200k": (("runs", 1200000, 5), [9]),
    "periodic_ab_100k": (("periodic", 100001, b"ab"), [9]),
    "periodic_long": (("periodic", 250000, b"the quick brown fox jumps over the lazy dog\n"), [9]),
    "zeros300k": (("bytes", b"\0" * 300000), [1, 9]),
}

# BWT.unbwtransform on (T, pidx) pairs that are NOT the BWT of anything (LF mapping with several cycles, chains
# that run off the end): the reference still walks n steps (lib/BWT.js:359-362); id -> pidx values
UNBWT_CASES = {"banana": [0, 1, 2, 3, 6], "mary9": [1, 17, 200], "text1k": [1, 500, 1000], "ab500": [1, 2, 999],
               "bytes40": [1, 5000], "a1000": [1, 1000], "abc_tie": [0, 3, 6], "text100k": [77777]}

# inputs used for stage-level (BWT) vectors
BWT_CASES = ["sample0", "sample1", "sample3", "a1", "a4", "ab500", "abc_tie", "banana", "mary9",
             "bytes40", "text1k", "text100k", "periodic_ab_100k", "lcg99981", "runs300k"]


def fixture_path(name: str):
    for d in FIXTURE_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    return None


def case_input(cid: str):
    """Return the input bytes (np.uint8 array) for a case id, or None if it is a reference
    fixture that is not available on this machine."""
    b = CASES[cid][0]
    kind = b[0]
    if kind == "fixture":
        p = fixture_path(b[1])
        if p is None:
            return None
        return np.fromfile(p, dtype=np.uint8)
    if kind == "bytes":
        return np.frombuffer(b[1], dtype=np.uint8).copy()
    if kind == "all_bytes":
        return synth.all_bytes(b[1])
    if kind == "lcg":
        return synth.lcg_ascii(b[1], b[2])
    if kind == "lcg_plus":
        return _lcg_plus(b[1], b[2], b[3])
    if kind == "text":
        return synth.text_like(b[1], b[2])
    if kind == "runs":
        return synth.runs_mixed(b[1], b[2])
    if kind == "periodic":
        return synth.periodic(b[1], b[2])
    raise KeyError(kind)
