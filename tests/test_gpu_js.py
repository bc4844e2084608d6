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
    assert r["bwtc_roundtrip"] is True
    assert r["unbwt"] == "banana"                          # BWT.unbwtransform, lib/BWT.js:352-363
    assert r["stream_len"] > 30
    assert r["badlevel"] == "Invalid block size multiplier"
    assert r["devices"] >= 1 and r["multi_same"] is True     # configure({devices, blocksInFlight}) -> cjs_bz2_compress_multi


def test_reference_acceptance_suite_against_the_dropin():
    """The reference's OWN mocha files, unchanged - test/bwtest.js, suftest.js, huffman.js, bzip2-basic.js,
    bzip2-block.js, bzip2-table.js and the bzip2 / bwtc round trips of file.js (all levels x sample0..5) - with
    require('../') resolving to js/index.js (js/run_reference_tests.js: describe/it stand-in + resolution hook).
    The files are staged under oracle/_ref/reftests by __graft_entry__.build(); they are not in the repository."""
    rt = os.path.join(ROOT, "oracle", "_ref", "reftests")
    if shutil.which("node") is None or not os.path.exists(os.path.join(ROOT, "build", "compressjs_amd.node")):
        pytest.skip("node or the addon is not available on this box")
    assert os.path.exists(os.path.join(rt, "test", "bwtest.js")), \
        "reference test files not staged (run __graft_entry__.build() in the build container)"
    r = subprocess.run(["node", os.path.join(ROOT, "js", "run_reference_tests.js"), rt], cwd=ROOT, capture_output=True, timeout=1500)
    out = r.stdout.decode().strip().splitlines()
    assert out, r.stderr.decode()[-2000:]
    res = json.loads(out[-1])
    assert res["failed"] == 0 and r.returncode == 0, res["failures"][:10]
    assert res["files"] == 7 and res["passed"] >= 150, res                 # 120 file.js round trips + the stage tests


def test_cli_round_trip(tmp_path):
    """bin/compressjs-amd (the reference's bin/compressjs flags): -z at the default level 7, -d, -b."""
    import bz2
    if shutil.which("node") is None or not os.path.exists(os.path.join(ROOT, "build", "compressjs_amd.node")):
        pytest.skip("node or the addon is not available on this box")
    cli = os.path.join(ROOT, "bin", "compressjs-amd")
    src = tmp_path / "in.txt"
    data = (b"The quick brown fox jumps over the lazy dog. " * 30000)[:1_200_000]
    src.write_bytes(data)
    z = tmp_path / "out.bz2"
    subprocess.check_call(["node", cli, "-z", str(src), str(z)], cwd=ROOT, timeout=300)
    zb = z.read_bytes()
    assert zb[:4] == b"BZh7" and bz2.decompress(zb) == data          # default level 7 (bin/compressjs:58)
    back = subprocess.check_output(["node", cli, "-d", str(z)], cwd=ROOT, timeout=300)
    assert back == data
    blk = subprocess.check_output(["node", cli, "-d", "-b", "32", str(z)], cwd=ROOT, timeout=300)
    assert data.startswith(blk) and len(blk) > 600000
    r = subprocess.run(["node", cli, "-d", "-9", str(z)], cwd=ROOT, capture_output=True, timeout=60)
    assert r.returncode == 1 and b"Compression level has no effect when decompressing." in r.stderr
