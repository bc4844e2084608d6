"""Pins the CPU oracle (oracle/bz2_oracle.c) against vectors produced by the reference itself
(tests/golden/golden.json, made by tests/golden/make_golden.py under node 12).  CPU only."""
import hashlib

import numpy as np
import pytest

import cases
import oracle


def _sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def _keys(golden, kind):
    return sorted(k for k in golden if k.split(":")[1:2] == [kind])


def _input(cid):
    d = cases.case_input(cid)
    if d is None:
        pytest.skip("reference fixture %s not staged on this machine" % cid)
    return d


# known-answer vectors of the reference's own tests (test/bwtest.js:38-79)
BWT2_KATS = [
    (b"bcababa", b"cbbaaab", 5),
    (b"ABCDEFGHIJKLMNOPQRSTUVWXYZ", b"ZABCDEFGHIJKLMNOPQRSTUVWXY", 0),
    (b"ZYXWVUTSRQPONMLKJIHGFEDCBA", b"BCDEFGHIJKLMNOPQRSTUVWXYZA", 25),
    (b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES",
     b"TEXYDST.E.IXIXIXXSSMPPS.B..E.S.EUSFXDIIOIIIT", 29),
    # tie KATs measured on the reference
    (b"aaaa", b"aaaa", 3), (b"abab", b"bbaa", 1), (b"abcabc", b"ccaabb", 1),
    (b"banana", b"nnbaaa", 3), (b"a", b"a", 0), (b"ab", b"ba", 0), (b"ba", b"ba", 1),
]


@pytest.mark.parametrize("inp,out,idx", BWT2_KATS)
def test_bwt_cyclic_kat(inp, out, idx):
    u, p = oracle.bwt_cyclic(inp)
    assert u.tobytes() == out and p == idx


def test_bwt_linear_kat():
    u, p = oracle.bwt_linear(b"banana")
    assert u.tobytes() == b"annbaa" and p == 4


# test/huffman.js:15-77
FIB = [0, 1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765,
       10946, 17711, 28657, 46368, 75025, 121393, 196418, 317811, 514229, 832040, 1346269, 2178309,
       3524578, 5702887, 9227465, 14930352]
HUFF_KATS = [
    ([1], 32, [1]), ([1, 1], 32, [1, 1]), ([1] * 5, 32, [3, 3, 2, 2, 2]),
    ([0, 0, 1, 1, 1, 1], 3, [3, 3, 3, 3, 2, 2]),
    (FIB[:36], 20, [20] * 16 + [19, 19, 18, 17, 16, 16, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]),
    (FIB[:22], 20, [20, 20, 19, 19, 19, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]),
    (FIB[:21], 20, [20, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]),
    (FIB[:36], 6, [6] * 30 + [5, 5, 5, 4, 3, 2]),
]


@pytest.mark.parametrize("freq,maxlen,expect", HUFF_KATS)
def test_huffman_allocator_kat(freq, maxlen, expect):
    assert oracle.huff_lengths(freq, maxlen) == expect


def test_huffman_allocator_fuzz_vs_reference(golden):
    for c in golden["huff"]["cases"]:
        assert oracle.huff_lengths(c["freq"], c["max_len"]) == c["lengths"]


def test_bz2_stream_digests(golden):
    n = 0
    for k in _keys(golden, "bz2"):
        cid, _, lv = k.split(":")
        d = cases.case_input(cid)
        if d is None:
            continue
        v = golden[k]
        assert _sha(d) == v["in_sha256"], "input generator drifted for " + cid
        o = oracle.bz2_compress(d, int(lv))
        assert len(o) == v["out_len"], k
        assert _sha(o) == v["out_sha256"], k
        if "out_hex" in v:
            assert o.hex() == v["out_hex"], k
        n += 1
    assert n >= 30


def test_bz2_decodes_with_independent_decoder():
    import bz2
    for cid in ("text100k", "runs300k", "lcg250000", "a259", "bytes40"):
        d = cases.case_input(cid)
        for lv in (1, 9):
            assert bz2.decompress(oracle.bz2_compress(d, lv)) == d.tobytes()


def test_invalid_level():
    with pytest.raises(ValueError):
        oracle.bz2_compress(b"x", 0)
    with pytest.raises(ValueError):
        oracle.bz2_compress(b"x", 10)


def test_stage_vectors(golden):
    for k in _keys(golden, "crc"):
        d = cases.case_input(k.split(":")[0])
        if d is not None:
            assert oracle.crc32(d) == golden[k]["crc"], k
    for k in _keys(golden, "bwt2"):
        d = cases.case_input(k.split(":")[0])
        if d is None:
            continue
        u, p = oracle.bwt_cyclic(d)
        assert p == golden[k]["pidx"] and _sha(u) == golden[k]["u_sha256"], k
    for k in _keys(golden, "bwt"):
        d = cases.case_input(k.split(":")[0])
        if d is None:
            continue
        u, p = oracle.bwt_linear(d)
        assert p == golden[k]["pidx"] and _sha(u) == golden[k]["u_sha256"], k
        assert (oracle.unbwt_linear(u, p) == d).all()
    n_unbwt = 0
    for k in [k for k in golden if k.split(":")[1:2] == ["unbwt"]]:      # (T, pidx) pairs that are no BWT at all: lib/BWT.js:359-362
        cid, _, pidx = k.split(":")
        d = cases.case_input(cid)
        assert _sha(oracle.unbwt_linear(d, int(pidx))) == golden[k]["u_sha256"], k
        n_unbwt += 1
    assert n_unbwt == sum(len(v) for v in cases.UNBWT_CASES.values())
    for k in _keys(golden, "sa"):
        d = cases.case_input(k.split(":")[0])
        if d is None:
            continue
        assert _sha(oracle.suffixsort(d).astype("<i4").tobytes()) == golden[k]["sa_sha256"], k


def test_block_stage_dump_is_consistent():
    d = cases.case_input("text2500k")
    blocks = list(oracle.block_stages(d, 9))
    assert [b["in_len"] for b in blocks] == [899908, 899926, 700166]
    assert sum(b["in_len"] for b in blocks) == d.size
    for b in blocks:
        u, p = oracle.bwt_cyclic(b["T"])
        assert p == b["pidx"] and (u == b["U"]).all()
        assert b["A"][-1] == b["alphabet_size"] + 1
