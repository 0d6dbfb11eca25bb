"""`ebwt2InDel -G n`: the executable itself on n GPUs -- one process, one host thread per GPU, RCCL communicators
(ncclCommInitAll), input read once, sharded traversal + ncclAllReduce, sharded scan + ncclSend/ncclRecv of the compact
call-set images to GPU 0.  On the one-GPU box: the RCCL path with one rank, and the same thread / sharding / writer
logic with 2, 3 and 5 ranks sharing the GPU through EB_MULTI_BACKEND=host (the two collectives replaced by exchanges
through host memory; RCCL refuses two ranks on one device).  Output: the reference-written golden .snp, byte for byte."""
import os
import subprocess

import pytest

from ebwt2indel_amd import _lib
from tests.conftest import CASES, golden_bytes

pytestmark = pytest.mark.gpu


def _case(name):
    return [c for c in CASES if c["name"] == name][0]


def _run(tmp_path, case, gpus, env_extra=None):
    inp = case["inputs"]
    files = {}
    for key, fname in inp.items():
        pth = tmp_path / key
        pth.write_bytes(golden_bytes(fname))
        files[key] = str(pth)
    out = tmp_path / f"out_{gpus}.snp"
    args = ["-1", files["bwt1"], "-o", str(out), "-G", str(gpus)] + (["-2", files["bwt2"]] if case["mode"] == 2 else []) + \
           (["-d", files["da"]] if case["mode"] == 3 else []) + list(case["args"])
    env = dict(os.environ, EB_NAV_BATCH=",,8", **(env_extra or {}))
    r = subprocess.run([_lib.CLI_PATH] + args, capture_output=True, text=True, timeout=600, env=env)
    return r, out


@pytest.mark.parametrize("records", ["own_link", "rccl_gather"])
@pytest.mark.parametrize("name", ["m1_default", "m1_edge_m2", "m2_het_default", "m3_hom_default", "m1_tiny_nocluster"])
def test_cli_G1_single_rank(tmp_path, name, records):
    """-G 1: the multi-GPU code path with one rank (no communicator needed) -- shard-form flags + finalize, device-only job with its
    wire image, the image downloaded over the rank's own link into the page-locked arena (default) or taken the round-4 way
    (EB_MULTI_BACKEND=rccl-gather: the images meet on GPU 0, which downloads them), unpack, writer: the golden .snp either way."""
    case = _case(name)
    r, out = _run(tmp_path, case, 1, {"EB_MULTI_BACKEND": "rccl-gather"} if records == "rccl_gather" else None)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_bytes() == golden_bytes(case["snp"])
    assert "on 1 GPU(s) [one rank: no communicator]" in r.stdout


@pytest.mark.parametrize("name", ["m1_default", "m2_het_default", "m3_het_default", "m2_edge_default"])
@pytest.mark.parametrize("gpus", [2, 3, 5])
def test_cli_G_ranks_share_the_gpu_through_host_exchange(tmp_path, name, gpus):
    case = _case(name)
    r, out = _run(tmp_path, case, gpus, {"EB_MULTI_BACKEND": "host"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_bytes() == golden_bytes(case["snp"])
    assert f"on {gpus} GPU(s) [host-memory exchange" in r.stdout
    one, out1 = _run(tmp_path, case, 1)
    # the counters the reference prints add up over the ranks
    pick = lambda log: [l for l in log.splitlines() if l.startswith(("Processed", "Found", "Computed", "Analyzed", "Stored"))]
    assert pick(r.stdout) == pick(one.stdout)


@pytest.mark.parametrize("name", ["m1_default", "m2_het_default", "m3_hom_default"])
@pytest.mark.parametrize("complement", ["1", "0"])
def test_cli_G1_host_exchange_single_rank(tmp_path, name, complement):
    """ONE rank through the host-exchange path: a one-shard build must come back in shard form like any other, or
    eb_flags_import_shard_sum inverts the threshold vectors a second time (round-2 advisor finding).  EB_NAV_FORM=bytes
    (the reference's form for the plain build) must not change what a shard hands out."""
    case = _case(name)
    r, out = _run(tmp_path, case, 1, {"EB_MULTI_BACKEND": "host", **({"EB_NAV_FORM": "bytes"} if complement == "0" else {})})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_bytes() == golden_bytes(case["snp"])
    assert "on 1 GPU(s) [host-memory exchange" in r.stdout


@pytest.mark.parametrize("name", ["m1_default", "m3_het_default"])
def test_cli_G8_ranks_share_the_gpu_through_host_exchange(tmp_path, name):
    """EIGHT ranks -- the node north_star names -- as eight host threads of one process on the one GPU: sharded traversal,
    sharded scan, call sets gathered in rank order, the N = 1 .snp and counters."""
    case = _case(name)
    r, out = _run(tmp_path, case, 8, {"EB_MULTI_BACKEND": "host"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_bytes() == golden_bytes(case["snp"])
    assert "on 8 GPU(s) [host-memory exchange" in r.stdout
    one, out1 = _run(tmp_path, case, 1)
    pick = lambda log: [l for l in log.splitlines() if l.startswith(("Processed", "Found", "Computed", "Analyzed", "Stored"))]
    assert pick(r.stdout) == pick(one.stdout)


def test_cli_G_needs_one_gpu_per_rank_and_reports_bad_input(tmp_path):
    import torch
    case = _case("m1_default")
    if torch.cuda.device_count() < 2:
        r, _ = _run(tmp_path, case, 2)
        assert r.returncode == 1 and "RCCL needs one GPU per rank" in r.stdout
    bad = bytearray(golden_bytes("tiny.ebwt"))
    bad[3] = ord("N")
    (tmp_path / "bad.ebwt").write_bytes(bytes(bad))
    for extra in ({}, {"EB_MULTI_BACKEND": "host"}):
        g = "1" if not extra else "3"
        r = subprocess.run([_lib.CLI_PATH, "-1", str(tmp_path / "bad.ebwt"), "-o", str(tmp_path / "o.snp"), "-G", g], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **extra))
        assert r.returncode == 1 and "read forbidden character 'N' (ASCII code 78)" in r.stdout
