"""filter_snp: the repository's post-filter against outputs of the reference's filter_snp binary."""
import json
import os
import subprocess

import pytest

from ebwt2indel_amd import _lib
from tests.conftest import GOLD, golden_bytes

TOOL = os.path.join(os.path.dirname(_lib.LIB_PATH), "filter_snp")
with open(os.path.join(GOLD, "manifest.json")) as f:
    FILTER_CASES = json.load(f)["filter_snp"]


@pytest.mark.parametrize("case", FILTER_CASES, ids=lambda c: c["out"])
def test_filter_snp_matches_reference(tmp_path, case):
    src = tmp_path / "in.snp"
    src.write_bytes(golden_bytes(case["snp"]))
    got = subprocess.run([TOOL, str(src)] + case["args"], capture_output=True).stdout
    assert got == golden_bytes(case["out"])
    assert got.count(b">") == case["records"]


def test_filter_snp_usage():
    r = subprocess.run([TOOL], capture_output=True, text=True)
    assert r.returncode == 0 and "filter_snp calls.snp m [M]" in r.stdout
