#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the BUILD CONTAINER only).

Inputs (.ebwt / .da) come from the repository's own simulator + suffix-sorting
eBWT builder (ebwt2indel_amd/sim.py); expected outputs (.snp) are written by the
REFERENCE BINARY oracle/_ref/ebwt2InDel_ref, compiled from the upstream sources by
oracle/Makefile.  Fixtures are data only (inputs + expected outputs); nothing of the
reference's source travels.  Everything is gzip-compressed; manifest.json lists the
cases (mode, input files, CLI arguments, expected output, reference exit status).

Usage:  python tests/gen_golden.py        (rewrites tests/golden/)
"""
import gzip
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

GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(ROOT, "oracle", "_ref", "ebwt2InDel_ref")

BASES = sim.BASES


def put(name, data: bytes):
    with gzip.GzipFile(os.path.join(GOLD, name + ".gz"), "wb", mtime=0) as f:
        f.write(data)


def run_ref(args, outpath):
    r = subprocess.run([REF] + args + ["-o", outpath], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return r.returncode


def variable_reads(haps, n_reads, lo, hi, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for j in range(n_reads):
        hap = haps[j % len(haps)]
        ln = int(rng.integers(lo, hi + 1))
        s = int(rng.integers(0, len(hap) - ln + 1))
        out.append(hap[s:s + ln].tobytes())
    return out


def edge_haplotypes(seed):
    """Three haplotypes over a 6 kb genome: tri-allelic sites, a SNP in front of a poly-A
    stretch (low-complexity right context), SNPs close together (max_snvs filter), 1..12 base
    indels (max_gap filter), and a variant 10 bases from the genome start (short left context)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = BASES[rng.integers(0, 4, size=6000, dtype=np.uint8)].copy()
    g[2000:2040] = ord("A")          # poly-A: right context of the SNP at 1999 starts with a run
    g[1999] = ord("C")
    h2 = g.copy()
    h3 = g.copy()

    def other(c, k):
        return BASES[(int(np.where(BASES == c)[0][0]) + k) % 4]

    for p in (10, 300, 700, 1999, 2500, 2503, 2506, 2509, 3000, 3400):   # 2500..2509: four SNPs within 10 bases
        h2[p] = other(g[p], 1)
    for p in (700, 3000):                                                 # third allele
        h3[p] = other(g[p], 2)
    h2 = h2.tolist()
    # indels on haplotype 2 (applied right-to-left so coordinates stay valid)
    for p, kind, ln in sorted([(5200, "D", 12), (4800, "I", 11), (4400, "D", 3), (4000, "I", 2), (3700, "D", 1)], reverse=True):
        if kind == "D":
            del h2[p:p + ln]
        else:
            h2[p:p] = BASES[rng.integers(0, 4, size=ln, dtype=np.uint8)].tolist()
    return g, np.array(h2, dtype=np.uint8), h3


def main():
    if not os.path.exists(REF):
        sys.exit("reference binary missing: run `make -C oracle ref` first")
    os.makedirs(GOLD, exist_ok=True)
    for f in os.listdir(GOLD):
        os.remove(os.path.join(GOLD, f))
    cases = []
    tmp = tempfile.mkdtemp(prefix="golden_")

    def add_case(name, mode, files, args_extra, term=35):
        """files: dict role -> (fixture name, bytes).  Runs the reference, stores expected .snp."""
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
        cases.append({
            "name": name, "mode": mode, "args": args_extra,
            "inputs": {role: fname for role, (fname, _) in files.items()},
            "snp": name + ".snp", "ref_exit": rc, "records": snp.count(b">"),
            "sha256": hashlib.sha256(snp).hexdigest(),
        })
        print(f"{name:28s} mode {mode} rc={rc} records={snp.count(b'>')}")

    # ---- mode 1: diploid sample, defaults and every option the CLI has
    reads, _, _ = sim.diploid_sample(12000, 2400, 100, seed=7, snp_every=400, indel_every=1500)
    e = sim.build_ebwt(reads).tobytes()
    f1 = {"bwt1": ("dip.ebwt", e)}
    add_case("m1_default", 1, f1, [])
    add_case("m1_m5", 1, f1, ["-m", "5"])
    add_case("m1_k20", 1, f1, ["-k", "20"])
    add_case("m1_L20_R25", 1, f1, ["-L", "20", "-R", "25"])
    add_case("m1_L40_R35", 1, f1, ["-L", "40", "-R", "35"])
    add_case("m1_g3", 1, f1, ["-g", "3"])
    add_case("m1_v1", 1, f1, ["-v", "1"])
    add_case("m1_c5", 1, f1, ["-c", "5"])
    add_case("m1_c2", 1, f1, ["-c", "2"])
    add_case("m1_g1", 1, f1, ["-g", "1"])
    add_case("m1_q2", 1, f1, ["-q", "2"])
    add_case("m1_m2_k12", 1, f1, ["-m", "2", "-k", "12"])

    # ---- mode 1: edge haplotypes (tri-allelic, poly-A, clustered SNPs, long indels), variable read length
    g, h2, h3 = edge_haplotypes(21)
    vreads = variable_reads([g, h2, h3], 3600, 40, 120, seed=22)
    ev = sim.build_ebwt(vreads).tobytes()
    f2 = {"bwt1": ("edge.ebwt", ev)}
    add_case("m1_edge_default", 1, f2, [])
    add_case("m1_edge_q1", 1, f2, ["-q", "1"])
    add_case("m1_edge_q2", 1, f2, ["-q", "2"])
    add_case("m1_edge_c45", 1, f2, ["-c", "45"])
    add_case("m1_edge_g12_v4", 1, f2, ["-g", "12", "-v", "4"])
    add_case("m1_edge_m2", 1, f2, ["-m", "2"])

    # ---- mode 1: n an exact multiple of 128 and a non-default terminator ('$' = 36)
    r128 = sim.sample_reads([g, h2], 1280, 99, seed=23)      # 1280 * 100 = 128000 = 1000 * 128
    e128 = sim.build_ebwt(r128, term=36).tobytes()
    assert len(e128) % 128 == 0
    add_case("m1_n128_term36", 1, {"bwt1": ("n128_t36.ebwt", e128)}, ["-t", "36"])

    # ---- modes 2 and 3: two clean homozygous samples (mode 2 == mode 3 there), then a heterozygous second sample
    r0, r1 = sim.two_samples(9000, 1800, 100, seed=11, snp_every=300, indel_every=1100)
    a = sim.build_ebwt(r0).tobytes()
    b = sim.build_ebwt(r1).tobytes()
    m, da = sim.build_merged_ebwt(r0, r1)
    f23 = {"bwt1": ("hom_a.ebwt", a), "bwt2": ("hom_b.ebwt", b)}
    f3 = {"bwt1": ("hom_m.ebwt", m.tobytes()), "da": ("hom_m.da", da.tobytes())}
    add_case("m2_hom_default", 2, f23, [])
    add_case("m2_hom_m4_v1", 2, f23, ["-m", "4", "-v", "1"])
    add_case("m2_hom_L25_R20_k14", 2, f23, ["-L", "25", "-R", "20", "-k", "14"])
    add_case("m3_hom_default", 3, f3, [])
    add_case("m3_hom_m4_v1", 3, f3, ["-m", "4", "-v", "1"])
    add_case("m3_hom_g2", 3, f3, ["-g", "2"])
    add_case("m3_hom_c2_g1", 3, f3, ["-c", "2", "-g", "1"])
    add_case("m2_hom_c2_g1", 2, f23, ["-c", "2", "-g", "1"])

    r0, r1 = sim.two_samples(9000, 1800, 100, seed=12, het_second=True, snp_every=300, indel_every=1100)
    r1 = r1[:1500]                                           # collections of unequal size
    a = sim.build_ebwt(r0).tobytes()
    b = sim.build_ebwt(r1).tobytes()
    m, da = sim.build_merged_ebwt(r0, r1)
    g23 = {"bwt1": ("het_a.ebwt", a), "bwt2": ("het_b.ebwt", b)}
    g3 = {"bwt1": ("het_m.ebwt", m.tobytes()), "da": ("het_m.da", da.tobytes())}
    add_case("m2_het_default", 2, g23, [])
    add_case("m2_het_q1", 2, g23, ["-q", "1"])
    add_case("m3_het_default", 3, g3, [])
    add_case("m3_het_q1", 3, g3, ["-q", "1"])

    # ---- modes 2/3 on the edge haplotypes (sample 0 = genome, sample 1 = h2 + h3), variable read length
    v0 = variable_reads([g], 1500, 50, 110, seed=31)
    v1 = variable_reads([h2, h3], 1700, 50, 110, seed=32)
    a = sim.build_ebwt(v0).tobytes()
    b = sim.build_ebwt(v1).tobytes()
    m, da = sim.build_merged_ebwt(v0, v1)
    add_case("m2_edge_default", 2, {"bwt1": ("edge_a.ebwt", a), "bwt2": ("edge_b.ebwt", b)}, [])
    add_case("m3_edge_default", 3, {"bwt1": ("edge_m.ebwt", m.tobytes()), "da": ("edge_m.da", da.tobytes())}, [])

    # ---- cluster-free input: the reference writes an empty .snp, then dies with SIGFPE
    tiny = sim.build_ebwt([b"ACGTACGGTCA", b"TTGACCA"]).tobytes()
    add_case("m1_tiny_nocluster", 1, {"bwt1": ("tiny.ebwt", tiny)}, [])

    # ---- filter_snp (post-filter of the one-sample pipeline): reference outputs on two of the .snp files above
    filt = []
    FREF = os.path.join(ROOT, "oracle", "_ref", "filter_snp_ref")
    for src, fargs in (("m1_default", ["5"]), ("m1_default", ["3", "8"]), ("m3_het_default", ["4", "12"]), ("m1_edge_default", ["1"])):
        snp_path = os.path.join(tmp, src + ".snp")
        out = subprocess.run([FREF, snp_path] + fargs, capture_output=True).stdout
        name = f"filter_{src}_{'_'.join(fargs)}.txt"
        put(name, out)
        filt.append({"snp": src + ".snp", "args": fargs, "out": name, "records": out.count(b">")})
        print(f"{name:40s} records={out.count(b'>')}")

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump({"generator": "tests/gen_golden.py", "reference": "oracle/_ref/ebwt2InDel_ref (upstream Release flags)",
                   "cases": cases, "filter_snp": filt}, f, indent=1)
    total = sum(os.path.getsize(os.path.join(GOLD, x)) for x in os.listdir(GOLD))
    print(f"{len(cases)} cases, {total / 1e6:.2f} MB in {GOLD}")


if __name__ == "__main__":
    main()
