"""Randomised differential run: small inputs of random shape (all three modes, reads with / without errors and reverse complements,
random options) through the GPU path and through the CPU oracle; every .snp must be the oracle's byte for byte.  Test tooling (it runs
the oracle): tests/test_gpu_fuzz.py runs a short seeded slice in the suite; `python -m tests.fuzz_vs_oracle <first seed> <count>` runs more."""
import sys

import numpy as np

from ebwt2indel_amd import api, sim
from tests import _oracle


def one_case(orc, seed: int):
    rng = np.random.Generator(np.random.PCG64(seed))
    mode = int(rng.integers(1, 4))
    L = int(rng.integers(30, 62)) if rng.integers(0, 4) == 0 else int(rng.integers(62, 131))   # (the default contexts need reads of 62 bases)
    n_reads = int(rng.integers(300, 6000))
    cov = float(rng.choice([8.0, 15.0, 30.0, 60.0]))
    err = int(rng.choice([0, 0, 5_000, 20_000]))
    rc = bool(rng.integers(0, 2)) and mode == 1
    glen = max(2 * L, int(n_reads * L / cov))
    args = []
    if rng.integers(0, 2):
        args += ["-m", str(int(rng.integers(1, 7)))]
    if rng.integers(0, 2):
        args += ["-k", str(int(rng.integers(6, 22)))]
    if rng.integers(0, 3) == 0:
        lft = int(rng.integers(8, min(L, 72)))
        args += ["-L", str(lft), "-R", str(int(rng.integers(6, lft + 1)))]
    if rng.integers(0, 3) == 0:
        args += ["-g", str(int(rng.integers(1, 12)))]
    if rng.integers(0, 3) == 0:
        args += ["-v", str(int(rng.integers(1, 4)))]
    if rng.integers(0, 4) == 0:
        args += ["-q", str(int(rng.integers(1, 4)))]
    if rng.integers(0, 4) == 0:
        args += ["-c", str(int(rng.integers(4, 30)))]
    po, pg = _oracle.Params.from_args(args), api.make_params(args)
    snp_every, indel_every = int(rng.integers(60, 600)), int(rng.integers(200, 2500))
    if mode == 1:
        reads, _, _ = sim.diploid_sample(glen, n_reads, L, seed=seed, snp_every=snp_every, indel_every=indel_every)
        if err:
            reads = sim.add_substitution_errors(reads, seed, err)
        if rc:
            reads = np.concatenate([reads, sim.reverse_complement(reads)])
        e = sim.build_ebwt(reads)
        want, _ = orc.run(e, params=po)
        got = api.run_one_dataset(e, pg)
    else:
        r0, r1 = sim.two_samples(glen, max(150, n_reads // 2), L, seed=seed, het_second=bool(rng.integers(0, 2)), snp_every=snp_every, indel_every=indel_every)
        if err:
            r0, r1 = sim.add_substitution_errors(r0, seed, err), sim.add_substitution_errors(r1, seed + 1, err)
        if mode == 2:
            a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
            want, _ = orc.run(a, bwt2=b, params=po)
            got = api.run_two_datasets(a, b, pg)
        else:
            m, da = sim.build_merged_ebwt(r0, r1)
            want, _ = orc.run(m, da=da, params=po)
            got = api.run_two_datasets_da(m, da, pg)
    return dict(seed=seed, mode=mode, read_len=L, reads=n_reads, errors_ppm=err, rc=rc, args=args, records=want.count(b">"), same=got == want)


def medium_case(orc, seed: int):
    """The same at 2-20 M positions (several scan tiles per workgroup, many batches of the ordered hand-over, launches of the traversal
    with full stacks): inputs from the GPU generator, the oracle single-threaded (seconds per case)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mode = int(rng.integers(1, 4))
    L = int(rng.integers(90, 151))
    n_reads = int(rng.integers(20_000, 130_000))
    cov = float(rng.choice([20.0, 30.0, 45.0]))
    err = int(rng.choice([0, 10_000])) if mode == 1 else 0
    rc = bool(rng.integers(0, 2)) and mode == 1 and err > 0
    args = []
    if rng.integers(0, 2):
        args += ["-m", str(int(rng.integers(2, 6)))]
    if rng.integers(0, 2):
        args += ["-k", str(int(rng.integers(12, 19)))]
    if rng.integers(0, 3) == 0:
        args += ["-g", str(int(rng.integers(2, 12)))]
    if rng.integers(0, 3) == 0:
        args += ["-v", str(int(rng.integers(1, 4)))]
    po, pg = _oracle.Params.from_args(args), api.make_params(args)
    if mode == 1:
        glen = int(n_reads * (2 if rc else 1) * L / cov)
        g = sim.random_genome(glen, seed)
        h2 = sim.mutate_haplotype(g, seed)
        text, offs = sim.haplotype_text([g, h2])
        starts = sim.sample_read_starts([g, h2], offs, n_reads, L, seed)
        if err:
            d_reads, d_starts = sim.error_sample_gpu(text, starts, L, seed, err, rc=rc)
            dev = sim.build_ebwt_gpu_slabbed(int(d_reads.data_ptr()), int(d_starts.data_ptr()), L, n_reads=int(d_starts.numel()), text_len=int(d_reads.numel()))
            del d_reads, d_starts
        else:
            dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
        e, _ = dev.to_host()
        dev.free()
        want, _ = orc.run(e, params=po)
        got = api.run_one_dataset(e, pg)
        n = len(e)
    else:
        text, s0, s1 = sim.two_sample_text(n_reads // 2, L, cov, genome_seed=seed, starts_seed=seed + 7)
        if mode == 2:
            da_, db_ = sim.build_ebwt_gpu_slabbed(text, s0, L), sim.build_ebwt_gpu_slabbed(text, s1, L)
            a, _ = da_.to_host()
            b, _ = db_.to_host()
            da_.free()
            db_.free()
            want, _ = orc.run(a, bwt2=b, params=po)
            got = api.run_two_datasets(a, b, pg)
            n = len(a) + len(b)
        else:
            dev = sim.build_ebwt_gpu_slabbed(text, np.concatenate([s0, s1]), L, n_first=len(s0), want_da=True)
            m, da = dev.to_host()
            dev.free()
            want, _ = orc.run(m, da=da, params=po)
            got = api.run_two_datasets_da(m, da, pg)
            n = len(m)
    return dict(seed=seed, mode=mode, n=n, read_len=L, errors_ppm=err, rc=rc, args=args, records=want.count(b">"), same=got == want)


def main(first: int, count: int, medium: bool = False) -> int:
    orc = _oracle.Oracle()
    bad, recs, by_mode = [], 0, {1: 0, 2: 0, 3: 0}
    for seed in range(first, first + count):
        r = medium_case(orc, seed) if medium else one_case(orc, seed)
        recs += r["records"]
        by_mode[r["mode"]] += 1
        if not r["same"]:
            bad.append(r)
            print("DIFFERS:", r, flush=True)
        if (seed - first + 1) % 25 == 0:
            print(f"  ... {seed - first + 1} cases, {recs} records, {len(bad)} differ", flush=True)
    print(f"{count} {'medium-size ' if medium else ''}cases from seed {first} (modes 1/2/3: {by_mode[1]}/{by_mode[2]}/{by_mode[3]}), {recs} records in all: {len(bad)} differ", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 100, medium=len(sys.argv) > 3 and sys.argv[3] == "medium"))
