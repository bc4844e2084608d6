"""The Node.js drop-in (js/index.js -> N-API addon -> C ABI -> HIP) against the golden digests."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_node_dropin_digests(golden):
    if shutil.which("node") is None or not os.path.exists(os.path.join(ROOT, "build", "compressjs_amd.node")):
        pytest.skip("node or the addon is not available on this box")
    out = subprocess.check_output(["node", os.path.join(ROOT, "js", "selftest.js")], cwd=ROOT, timeout=300)
    r = json.loads(out.decode().strip().splitlines()[-1])
    assert r["a1000"] == golden["a1000:bz2:9"]["out_sha256"]
    assert r["empty"] == golden["empty:bz2:9"]["out_sha256"]
    assert r["bytes40"] == golden["bytes40:bz2:9"]["out_sha256"]
    assert r["lcg250000"] == golden["lcg250000:bz2:1"]["out_sha256"]
    assert r["bwt"] == [5, "cbbaaab"]                       # test/bwtest.js:39-44
    assert r["bwtc_a1000"] == golden["a1000:bwtc:9"]["out_sha256"]
    assert r["bwtc_bytes40"] == golden["bytes40:bwtc:6"]["out_sha256"]
    assert r["bwt_linear"] == [4, "annbaa"]
    assert r["sa"] == [5, 3, 1, 0, 4, 2]
    assert r["huff"] == [3, 3, 2, 2, 2]                     # test/huffman.js:24-28
    assert r["roundtrip"] is True and r["block1"] is True and r["sized"] == 11   # GPU decoder behind Bzip2.decompressFile/Block
    assert r["table"] == golden["lcg250000:bz2:1"]["blocks"]                       # Bzip2.table == the reference's own table
    assert r["badcrc"] == ["TypeError", -5, "Data error: Bad stream CRC ()"]
    assert r["badmagic"] == [-2, "Not bzip data: bad magic"]
    assert r["unbwt"] == "banana"                          # BWT.unbwtransform, lib/BWT.js:352-363
    assert r["stream_len"] > 30
    assert r["badlevel"] == "Invalid block size multiplier"
