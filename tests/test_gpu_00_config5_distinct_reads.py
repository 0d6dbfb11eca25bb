"""BASELINE configs[4] -- mode 1, 1 G x 150 bp DISTINCT reads, 151 G BWT positions, HBM-resident -- on the one GPU of the test box.

Runs tests/config5_distinct_reads.py in a process of its own (it needs 250 of the 288 GB of HBM and ~190 GB of host memory, so it runs
FIRST, before this process holds anything on the device): the sample is made on the device, its eBWT by the slab-by-slab
generator, then index -> traversal -> phase 4, every stage against a checker --
  index       F and 100 k parallel_rank queries = the CPU oracle's (its index streamed off the GPU in pieces); parallel_rank =
              prefix counts of the ASCII symbols around all 35 superblock boundaries; 200 k reads spelled back through access + LF;
  traversal   every LCP slot written once; thrK / thrR = LCP >= K / k_right BY DEFINITION over all 151 G positions, minima = the
              LCP's local minima (the generator's own statement of the vectors, held to the oracle on small inputs in
              test_gpu_generator.py); AND the CPU oracle's own traversal -- the top of both trees in full, then one subtree in 256 below
              a frontier of 2^16 -- finds every LCP / minimum slot it would write as the GPU wrote it (round 5);
  phase 4     the whole-range .snp BYTE-IDENTICAL to the CPU oracle's whole-range pass (512 shards on the host threads, ~95 s) and, record
              by record with contexts, on 49 windows of 2 M positions (every superblock boundary, both ends, random places); the
              whole-range .snp = 8 shards' (the N = 8 ownership rule).
The script asserts all of that; here its figures are also held to the record of the builder's run (profiles/)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_one_billion_distinct_reads_whole_pipeline(tmp_path):
    import torch
    free_b, total_b = torch.cuda.mem_get_info(0)
    assert total_b > 280e9, "this test is sized for the 288 GB of an MI355X"
    torch.cuda.empty_cache()
    del torch
    with open(os.path.join(ROOT, "profiles", "r03_151G_distinct_reads.json")) as f:
        want = json.load(f)
    out = tmp_path / "distinct.json"
    env = dict(os.environ, EB_DISTINCT_OUT=str(out))
    r = subprocess.run([sys.executable, "-m", "tests.config5_distinct_reads"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1100)
    tail = r.stdout.decode(errors="replace")[-3000:]
    assert r.returncode == 0, tail
    with open(out) as f:
        got = json.load(f)
    for k in ("n", "text_symbols", "nodes", "leaves", "minima", "thrK_ones", "thrR_ones", "clusters_analysed", "call_set", "snp_records", "snp_sha256", "oracle_snp_sha256",
              "phase4_oracle_windows", "window_clusters", "window_records", "reads_spelled_back", "rank_positions_checked_against_definition"):
        assert got[k] == want[k], (k, got[k], want[k])
    assert got["n"] == 151_000_000_000 and got["every_lcp_slot_written_once"] and got["thrK_thrR_equal_definition"] and got["shards8_equal_whole"] and got["snp_equals_oracle_whole_range"]
    assert got["snp_sha256"] == got["oracle_snp_sha256"]
    smp = got["oracle_traversal_sample"]          # the reference's traversal itself (a sample of its subtrees + the whole top) against the GPU's vectors
    assert smp["mismatches"] == 0 and smp["nodes"] > got["nodes"] // (4 * smp["every"]) and smp["lcp_slots"] > got["n"] // (4 * smp["every"])
