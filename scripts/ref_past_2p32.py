"""The UNMODIFIED reference binary on an input past 2^32 BWT positions (build container, CPU only, ~20 min).

Every BASELINE config from [1] on lives beyond the first 2^32-symbol superblock (dna_string.hpp:36-38), where a plain
read collection of that size would keep the single-threaded reference busy for the better part of an hour.  The input
here is the eBWT of a diploid sample in which every read occurs COPIES times in a row: the eBWT is the base sample's
with each symbol repeated COPIES times (tests/test_sim.py), the suffix tree keeps the base sample's right-maximal
nodes, and the reference gets through 4.35 G positions in minutes.  Same input on the GPU box: sim.replicated_ebwt_gpu
(GPU generator + repeat_interleave); tests/test_gpu_superblocks.py requires the GPU path's .snp to have the SHA-256
recorded here (and the input's, so that both sides provably saw the same bytes).

    python scripts/ref_past_2p32.py            # writes profiles/r02_reference_past_2p32.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ebwt2indel_amd import sim  # noqa: E402

# the case (tests/test_gpu_superblocks.py uses the same constants)
N_READS, READ_LEN, COVERAGE, COPIES, SEED = 3_600_000, 150, 4.0, 8, 2
REF = os.path.join(ROOT, "oracle", "_ref", "ebwt2InDel_ref")
WORK = os.environ.get("EB_WORK", "/tmp/eb_ref_past_2p32")


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 26)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    os.makedirs(WORK, exist_ok=True)
    t0 = time.time()
    genome_len = int(N_READS * READ_LEN / COVERAGE)
    g = sim.random_genome(genome_len, SEED)
    h2 = sim.mutate_haplotype(g, SEED)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, N_READS, READ_LEN, SEED)
    reads = sim.reads_from_starts(text, starts, READ_LEN)
    del text, g, h2
    base = sim.build_ebwt(reads, threads=os.cpu_count() or 8)
    del reads
    print(f"base eBWT: n0={len(base)} in {time.time() - t0:.0f}s", flush=True)
    base_sha = hashlib.sha256(base.tobytes()).hexdigest()
    path, out = os.path.join(WORK, "x.ebwt"), os.path.join(WORK, "ref.snp")
    n = len(base) * COPIES
    with open(path, "wb") as f:
        step = 1 << 24
        for a in range(0, len(base), step):
            np.repeat(base[a:a + step], COPIES).tofile(f)
    del base
    in_sha = sha_file(path)
    assert os.path.getsize(path) == n and n > (1 << 32)
    print(f"input: n={n} ({n - (1 << 32)} positions past 2^32), sha256 {in_sha}", flush=True)
    t1 = time.time()
    r = subprocess.run([REF, "-1", path, "-o", out], capture_output=True, text=True)
    t_ref = time.time() - t1
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    snp_sha = sha_file(out)
    with open(out, "rb") as f:
        recs = sum(chunk.count(b">") for chunk in iter(lambda: f.read(1 << 26), b""))
    log = r.stdout

    def grab(pat):
        m = re.search(pat, log)
        return int(m.group(1)) if m else None
    res = {
        "what": "unmodified reference binary (oracle/_ref/ebwt2InDel_ref, g++ -Ofast of the upstream sources) on an eBWT past 2^32 positions",
        "input": {"n_reads_base": N_READS, "read_len": READ_LEN, "coverage_base": COVERAGE, "copies": COPIES, "seed": SEED,
                  "n": n, "positions_past_2p32": n - (1 << 32), "sha256": in_sha, "base_sha256": base_sha,
                  "generator": "sim.diploid_sample-style haplotypes + host suffix sorter here; sim.replicated_ebwt_gpu on the GPU box"},
        "options": "defaults (-L31 -R30 -k16 -g10 -v2 -m3 -c20 -q0)",
        "snp_sha256": snp_sha, "snp_bytes": os.path.getsize(out), "snp_records": recs,
        "reference_seconds": round(t_ref, 1), "host": "build container, 1 thread",
        "reference_counters": {"leaves": grab(r"Processed (\d+) suffix-tree leaves"), "nodes": grab(r"Processed (\d+) suffix-tree nodes"),
                               "minima": grab(r"Found (\d+) LCP minima"), "clusters": grab(r"Analyzed (\d+) clusters")},
        "reference_stdout_tail": [l for l in log.splitlines() if not re.match(r"^\d+\s", l) and "%" not in l][-25:],
    }
    dst = os.path.join(ROOT, "profiles", "r02_reference_past_2p32.json")
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("snp_sha256", "snp_records", "reference_seconds", "reference_counters")}), flush=True)
    if os.environ.get("EB_KEEP", "0") != "1":
        os.remove(path)


if __name__ == "__main__":
    main()
