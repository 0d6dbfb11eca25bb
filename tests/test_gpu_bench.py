"""bench.py end to end on the GPU at reduced size: the JSON contract, and the N=2 path (two ranks sharing the one
GPU of the test box, records gathered over gloo) producing the same .snp as N=1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--no-other-modes", "--reads", "300000", "--steps", "2", "--warmup", "1", "--rank-queries", "1048576", "--cpu-dram-reads", "0"]


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_contract_single_gpu():
    # (the DRAM-resident leg of the CPU baseline at a size that keeps the test short; the other tests here skip it)
    # (and with the short runs of modes 2 and 3 the default run appends: `other_modes`)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-reads", "20000"] + ARGS[1:-1] + ["300000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for m in ("mode2", "mode3"):
        o = d["other_modes"][m]
        assert "error" not in o, o
        assert o["snp_records"] > 500 and o["ms_per_step"] > 0 and o["algorithmic_bytes_per_position"] == 0.875 and f"mode {m[-1]}" in o["workload"]
    for m in ("errors_1pct", "errors_1pct_with_reverse_complements"):
        o = d["realistic_reads"][m]
        assert "error" not in o, o
        assert o["snp_records"] > 500 and o["ms_per_step"] > 0 and "substitution errors" in o["workload"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["snp_identical_to_gpu"]
    assert d["cpu_baseline"]["dram_resident"]["snp_identical_to_gpu"]
    assert d["snp_records"] > 1000
    pytest.single_gpu_sha = d["snp_sha256"]


@pytest.mark.parametrize("mode", [2, 3])
def test_bench_modes_2_and_3(mode):
    """`--mode 2` (two BWTs) and `--mode 3` (merged BWT + document array packed where the generator left it in HBM): the same line,
    its CPU leg run by the oracle and the reference binary in that mode on the same sample."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", str(mode), "--cpu-sample-reads", "20000"] + ARGS,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["config"]["mode"] == mode and "other_modes" not in d
    assert d["roofline"]["algorithmic_bytes_per_position"] == 0.875
    assert d["cpu_baseline"]["snp_identical_to_gpu"] and (d["cpu_baseline"]["reference"] is None or d["cpu_baseline"]["reference"]["snp_identical_to_gpu"])
    assert d["snp_records"] > 500
    if mode == 3:
        assert d["phases23_gpu"]["document_array_packed_from_hbm_seconds"] > 0


def test_bench_two_ranks_share_the_gpu_same_snp():
    env = dict(os.environ, EB_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-sample-reads", "0"] + ARGS
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-reads", "0"] + ARGS, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = _last_json(one.stdout)
    assert d["snp_sha256"] == d1["snp_sha256"] and d["snp_records"] == d1["snp_records"]
    assert d["clusters_emitted"] == d1["clusters_emitted"]


@pytest.mark.parametrize("gather", ["arena", "rccl"])
def test_bench_nccl_collectives_with_one_rank_same_snp(gather):
    """The N > 1 code path of the driver's multi-GPU run -- RCCL process group, sharded traversal with its all-reduce,
    device-only jobs with compact wire images, handed to rank 0 through the node's host arena (every rank's own PCIe link; the
    default) or gathered GPU to GPU with RCCL and resolved one step later (EB_BENCH_GATHER=rccl, rounds 2-4) -- exercised with a
    single rank (RCCL refuses two ranks on one device): same .snp as the plain N = 1 run either way."""
    env = dict(os.environ, EB_BENCH_FORCE_DIST="1", EB_BENCH_GATHER=gather, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-reads", "0", "--steps", "5"] + ARGS[:3] + ARGS[5:],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-reads", "0"] + ARGS, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = _last_json(one.stdout)
    assert d["snp_sha256"] == d1["snp_sha256"] and d["snp_records"] == d1["snp_records"]
    assert d["phases23_gpu"]["nodes"] == d1["phases23_gpu"]["nodes"] and d["phases23_gpu"]["rank_queries"] == d1["phases23_gpu"]["rank_queries"]


def test_bench_gpus_2_without_two_gpus_fails_loudly():
    """`python bench.py --gpus 2` started bare on a box with one GPU must not quietly run (and label) a 1-GPU job."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "needs 2 GPUs" in r.stderr


def test_bench_gpus_2_launches_its_own_ranks():
    """Bare `python bench.py --gpus 2` (no torchrun around it): the process starts the two ranks itself (gloo rehearsal:
    the ranks share the one GPU) and relays exactly one JSON line with n_gpus = 2 and the N = 1 .snp."""
    env = dict(os.environ, EB_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-sample-reads", "0"] + ARGS,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-sample-reads", "0"] + ARGS, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    assert d["snp_sha256"] == _last_json(one.stdout)["snp_sha256"]


def test_bench_refuses_mismatched_world_and_debug_switch():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"] + ARGS, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "refusing to mislabel" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + ARGS, capture_output=True, text=True, timeout=120, env=dict(os.environ, EB_SCAN_DEBUG="4"))
    assert r.returncode != 0 and "EB_SCAN_DEBUG" in r.stderr
