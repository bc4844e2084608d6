"""The CPU oracle's DECODER (oracle/bz2_decode_oracle.c) against what the reference's
Bzip2.decompressFile / decompressBlock / table did under node 12 (tests/golden/golden_decode.json)."""
import hashlib
import json
import os
import re

import pytest

import decode_cases
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gdec():
    with open(os.path.join(ROOT, "tests", "golden", "golden_decode.json")) as f:
        return json.load(f)["vectors"]


@pytest.fixture(scope="module")
def streams():
    return decode_cases.streams()


def expect_of(v):
    """golden record -> (ret or None, detail text or None)"""
    if v["ok"]:
        return v["out_len"], None
    m = re.match(r"^[^:]+(?:: (.*))?$", v["message"])
    detail = m.group(1) if m else None
    if detail and detail.startswith("Bad block CRC"):
        detail = "Bad block CRC"
    if detail and detail.startswith("Bad stream CRC"):
        detail = "Bad stream CRC"
    return v["error_code"], detail


def test_every_stream_vs_reference(gdec, streams):
    n = 0
    for sid, s, ms in streams:
        if s is None:
            continue
        v = gdec[sid]
        assert hashlib.sha256(s).hexdigest() == v["stream_sha256"], "stream recipe drifted: " + sid
        ret, det, out, tab = oracle.bz2_decompress(s, ms)
        want, wdet = expect_of(v)
        assert v["ok"] or v["error_type"] == "TypeError", sid      # the reference only throws its own errors here
        assert ret == want, (sid, ret, det, v)
        assert oracle.DECODE_DETAIL[det] == wdet, (sid, det, v)
        if v["ok"]:
            assert hashlib.sha256(out).hexdigest() == v["out_sha256"], sid
            if "table" in v:
                assert [list(t) for t in tab] == v["table"], sid
        n += 1
    assert n >= 100


def test_block_decode_vs_reference(gdec, streams):
    by = {sid: s for sid, s, _ in streams}
    n = 0
    for sid, bitpos in decode_cases.BLOCK_CASES:
        s = by.get(sid)
        if s is None:
            continue
        v = gdec["block:%s@%d" % (sid, bitpos)]
        ret, det, out = oracle.bz2_decompress_block(s, bitpos)
        want, wdet = expect_of(v)
        assert ret == want and oracle.DECODE_DETAIL[det] == wdet, (sid, bitpos, ret, det, v)
        if v["ok"]:
            assert hashlib.sha256(out).hexdigest() == v["out_sha256"]
        n += 1
    assert n >= 4
