#!/usr/bin/env python3
"""Round-5 additions to the golden fixtures (run in the BUILD CONTAINER only).

Same rules as tests/gen_golden.py / gen_golden_r04.py: inputs come from the repository's own simulator + suffix-sorting builder,
expected .snp files are written by the REFERENCE BINARY oracle/_ref/ebwt2InDel_ref (compiled from the upstream sources by
oracle/Makefile); fixtures are data only.  The cases are APPENDED to tests/golden/manifest.json.

  rc*       the reference's documented input (README.md:88-100): reads with 1 % substitution errors PLUS the reverse complement of
            every read, in all three modes -- every variant is then seen on both strands (two clusters per event)
  err -q 2 / short -L 20 ...   replacements for the two round-4 cases whose reference output is EMPTY (m1_err_m1_q1: -q 1 allows one
            variant per position and mode 1 prints from two; m1_short_default: 35-base reads never reach -L 31): these print.
            The empty ones stay as what they are -- empty-output cases ("records": 0 in the manifest).

Usage:  python tests/gen_golden_r05.py
"""
import gzip
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ebwt2indel_amd import sim  # noqa: E402
from tests.gen_golden import GOLD, REF, put, run_ref  # noqa: E402

TAG = "tests/gen_golden_r05.py"


def main():
    if not os.path.exists(REF):
        sys.exit("reference binary missing: run `make -C oracle ref` first")
    with open(os.path.join(GOLD, "manifest.json")) as f:
        man = json.load(f)
    man["cases"] = [c for c in man["cases"] if c.get("generator") != TAG]
    tmp = tempfile.mkdtemp(prefix="golden_r05_")
    cases = []

    def add_case(name, mode, files, args_extra):
        paths = {}
        for role, (fname, data) in files.items():
            paths[role] = os.path.join(tmp, fname)
            with open(paths[role], "wb") as f:
                f.write(data)
            if not os.path.exists(os.path.join(GOLD, fname + ".gz")):
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
        assert snp.count(b">") > 0, "a round-5 case must print"

    def gold(fname):
        with gzip.open(os.path.join(GOLD, fname + ".gz"), "rb") as f:
            return f.read()

    # ---- mode 1: errors + reverse complements (sim.error_sample: the recipe the full-size config2rc input is made with)
    g = sim.random_genome(12000, 51)
    h2 = sim.mutate_haplotype(g, 51, snp_every=350, indel_every=1300)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, 1800, 100, 51)
    reads = sim.error_sample(text, starts, 100, 51, 10_000, rc=True)
    e = sim.build_ebwt(reads).tobytes()
    f1 = {"bwt1": ("rc.ebwt", e)}
    add_case("m1_rc_default", 1, f1, [])
    add_case("m1_rc_m2", 1, f1, ["-m", "2"])
    add_case("m1_rc_k20_L40_R36", 1, f1, ["-k", "20", "-L", "40", "-R", "36"])

    # ---- modes 2 / 3: two individuals, each collection with errors and its reverse complements
    g = sim.random_genome(9000, 52)
    h2 = sim.mutate_haplotype(g, 52, snp_every=300, indel_every=1100)
    t0, o0 = sim.haplotype_text([g])
    t1, o1 = sim.haplotype_text([g, h2])
    r0 = sim.error_sample(t0, sim.sample_read_starts([g], o0, 1200, 100, 53), 100, 53, 10_000, rc=True)
    r1 = sim.error_sample(t1, sim.sample_read_starts([g, h2], o1, 1200, 100, 54), 100, 54, 10_000, rc=True)
    a, b = sim.build_ebwt(r0).tobytes(), sim.build_ebwt(r1).tobytes()
    m, da = sim.build_merged_ebwt(r0, r1)
    add_case("m2_rc_default", 2, {"bwt1": ("rc_a.ebwt", a), "bwt2": ("rc_b.ebwt", b)}, [])
    add_case("m3_rc_default", 3, {"bwt1": ("rc_m.ebwt", m.tobytes()), "da": ("rc_m.da", da.tobytes())}, [])
    add_case("m3_rc_m2_q2", 3, {"bwt1": ("rc_m.ebwt", m.tobytes()), "da": ("rc_m.da", da.tobytes())}, ["-m", "2", "-q", "2"])

    # ---- the round-4 inputs whose cases came out empty, with parameters that print
    add_case("m1_err_m1_q2", 1, {"bwt1": ("err.ebwt", gold("err.ebwt"))}, ["-m", "1", "-q", "2"])
    add_case("m1_short_L20_R12_k10", 1, {"bwt1": ("short.ebwt", gold("short.ebwt"))}, ["-L", "20", "-R", "12", "-k", "10"])

    man["cases"] += cases
    man["generator"] = ("tests/gen_golden.py; cases marked with \"generator\": tests/gen_golden_r04.py / tests/gen_golden_r05.py; "
                        "cases with \"records\": 0 are empty-output cases (the reference prints nothing; m1_tiny_nocluster also dies with SIGFPE after writing the empty file)")
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
    total = sum(os.path.getsize(os.path.join(GOLD, x)) for x in os.listdir(GOLD))
    print(f"+{len(cases)} cases ({len(man['cases'])} in all, {sum(1 for c in man['cases'] if c['records'] > 0)} of them print), {total / 1e6:.2f} MB in {GOLD}")


if __name__ == "__main__":
    main()
