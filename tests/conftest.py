import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["vectors"]


@pytest.fixture(scope="session")
def golden_big():
    """Digests of the reference's own output (node 12, /root/reference) on the full 10^8-byte bench streams
    (tests/golden/make_golden_big.py)."""
    with open(os.path.join(ROOT, "tests", "golden", "golden_big.json")) as f:
        return json.load(f)["vectors"]
