import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)["cases"]


def golden_bytes(name: str) -> bytes:
    with gzip.open(os.path.join(GOLD, name + ".gz"), "rb") as f:
        return f.read()


CASES = load_manifest()
CASE_IDS = [c["name"] for c in CASES]


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle
    return _oracle.Oracle()
