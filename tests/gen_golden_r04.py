#!/usr/bin/env python3
"""Round-4 additions to the golden fixtures (run in the BUILD CONTAINER only): messier inputs than tests/gen_golden.py's.

Same rules: inputs come from the repository's own simulator + suffix-sorting builder, expected .snp files are written by the
REFERENCE BINARY oracle/_ref/ebwt2InDel_ref (compiled from the upstream sources by oracle/Makefile); fixtures are data only.
The cases are APPENDED to tests/golden/manifest.json (existing entries and files are left alone; re-running replaces this
script's own cases):

  err*      reads with 1 % substitution errors (every error makes its own little clusters; letters no longer come in clean
            runs, which is what the FL table's void chunks, the scan's queues and the right-context gate see on real data)
  deep      80x coverage: clusters longer than the 64..128 rows the scan keeps in registers (the memory path of long runs)
  short     35-base reads with -L 31 -R 30: most left contexts run into a read start before they reach k_left
  uneq      mode 2 / 3 with the second collection three times the first (and the other way round)
  options   K above k_right, -m 1, contexts of three words (-L 70 -R 66), -k 5

Usage:  python tests/gen_golden_r04.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ebwt2indel_amd import sim  # noqa: E402
from tests.gen_golden import GOLD, REF, put, run_ref  # noqa: E402

TAG = "tests/gen_golden_r04.py"


def with_errors(reads, rate, seed):
    """Substitution errors: each base is replaced by one of the three others with probability `rate`."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for r in reads:
        a = np.frombuffer(bytes(r), dtype=np.uint8).copy()
        hit = np.flatnonzero(rng.random(len(a)) < rate)
        for p in hit:
            others = [c for c in b"ACGT" if c != a[p]]
            a[p] = others[int(rng.integers(0, 3))]
        out.append(a.tobytes())
    return out


def as_list(reads):
    return [bytes(r) for r in reads]


def main():
    if not os.path.exists(REF):
        sys.exit("reference binary missing: run `make -C oracle ref` first")
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    man["cases"] = [c for c in man["cases"] if c.get("generator") != TAG]
    tmp = tempfile.mkdtemp(prefix="golden_r04_")
    cases = []

    def add_case(name, mode, files, args_extra):
        paths = {}
        for role, (fname, data) in files.items():
            paths[role] = os.path.join(tmp, fname)
            with open(paths[role], "wb") as f:
                f.write(data)
            put(fname, data)
        args = ["-1", paths["bwt1"]]
        if mode == 2:
            args += ["-2", paths["bwt2"]]
        if mode == 3:
            args += ["-d", paths["da"]]
        args += args_extra
        out = os.path.join(tmp, name + ".snp")
        rc = run_ref(args, out)
        snp = open(out, "rb").read()
        put(name + ".snp", snp)
        cases.append({"name": name, "mode": mode, "args": args_extra, "inputs": {role: fname for role, (fname, _) in files.items()},
                      "snp": name + ".snp", "ref_exit": rc, "records": snp.count(b">"), "sha256": hashlib.sha256(snp).hexdigest(), "generator": TAG})
        print(f"{name:28s} mode {mode} rc={rc} records={snp.count(b'>')}")

    # ---- mode 1, reads with sequencing errors
    reads, _, _ = sim.diploid_sample(10000, 3000, 100, seed=41, snp_every=350, indel_every=1300)
    er = with_errors(as_list(reads), 0.01, 42)
    e = sim.build_ebwt(er).tobytes()
    f1 = {"bwt1": ("err.ebwt", e)}
    add_case("m1_err_default", 1, f1, [])
    add_case("m1_err_m2", 1, f1, ["-m", "2"])
    add_case("m1_err_m4_k14_q3", 1, f1, ["-m", "4", "-k", "14", "-q", "3"])

    # ---- modes 2 / 3, both collections with errors
    r0, r1 = sim.two_samples(8000, 2000, 100, seed=43, het_second=True, snp_every=300, indel_every=1100)
    e0, e1 = with_errors(as_list(r0), 0.01, 44), with_errors(as_list(r1), 0.01, 45)
    a, b = sim.build_ebwt(e0).tobytes(), sim.build_ebwt(e1).tobytes()
    m, da = sim.build_merged_ebwt(e0, e1)
    add_case("m2_err_default", 2, {"bwt1": ("err_a.ebwt", a), "bwt2": ("err_b.ebwt", b)}, [])
    add_case("m3_err_default", 3, {"bwt1": ("err_m.ebwt", m.tobytes()), "da": ("err_m.da", da.tobytes())}, [])
    add_case("m3_err_m2", 3, {"bwt1": ("err_m.ebwt", m.tobytes()), "da": ("err_m.da", da.tobytes())}, ["-m", "2"])

    # ---- 80x coverage: clusters of 100+ rows
    reads, _, _ = sim.diploid_sample(3000, 2400, 100, seed=46, snp_every=250, indel_every=900)
    d = sim.build_ebwt(reads).tobytes()
    add_case("m1_deep_default", 1, {"bwt1": ("deep.ebwt", d)}, [])
    add_case("m1_deep_m20", 1, {"bwt1": ("deep.ebwt", d)}, ["-m", "20"])

    # ---- reads barely longer than the contexts asked for
    reads, _, _ = sim.diploid_sample(4000, 4000, 35, seed=47, snp_every=200, indel_every=800)
    s = sim.build_ebwt(reads).tobytes()
    add_case("m1_short_default", 1, {"bwt1": ("short.ebwt", s)}, [])
    add_case("m1_short_L12_R10_k8", 1, {"bwt1": ("short.ebwt", s)}, ["-L", "12", "-R", "10", "-k", "8"])

    # ---- two collections of very different size, both orders
    g = sim.random_genome(9000, 48)
    h2 = sim.mutate_haplotype(g, 48, snp_every=300, indel_every=1100)
    u0, u1 = as_list(sim.sample_reads([g], 900, 100, 48)), as_list(sim.sample_reads([g, h2], 2700, 100, 49))
    a, b = sim.build_ebwt(u0).tobytes(), sim.build_ebwt(u1).tobytes()
    add_case("m2_uneq_small_large", 2, {"bwt1": ("uneq_a.ebwt", a), "bwt2": ("uneq_b.ebwt", b)}, [])
    add_case("m2_uneq_large_small", 2, {"bwt1": ("uneq_b.ebwt", b), "bwt2": ("uneq_a.ebwt", a)}, [])
    m, da = sim.build_merged_ebwt(u0, u1)
    add_case("m3_uneq", 3, {"bwt1": ("uneq_m.ebwt", m.tobytes()), "da": ("uneq_m.da", da.tobytes())}, [])

    # ---- option corners on the round-1 diploid sample (its input file is in the tree already): K above k_right, coverage threshold 1,
    # contexts of three 64-bit words, a K so small that clusters run over hundreds of rows
    import gzip
    with gzip.open(os.path.join(GOLD, "dip.ebwt.gz"), "rb") as f:
        dip = f.read()
    fd = {"bwt1": ("dip.ebwt", dip)}
    add_case("m1_k32_R20", 1, fd, ["-k", "32", "-R", "20"])
    add_case("m1_m1", 1, fd, ["-m", "1"])
    add_case("m1_L70_R66", 1, fd, ["-L", "70", "-R", "66"])
    add_case("m1_k5", 1, fd, ["-k", "5"])
    add_case("m1_err_m1_q1", 1, f1, ["-m", "1", "-q", "1"])

    man["cases"] += cases
    man["generator"] = "tests/gen_golden.py; cases marked with \"generator\": tests/gen_golden_r04.py"
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
    total = sum(os.path.getsize(os.path.join(GOLD, x)) for x in os.listdir(GOLD))
    print(f"+{len(cases)} cases ({len(man['cases'])} in all), {total / 1e6:.2f} MB in {GOLD}")


if __name__ == "__main__":
    main()
