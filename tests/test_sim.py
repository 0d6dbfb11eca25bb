"""Synthetic-input tooling: the C++ suffix-sorting builder against the pure-Python definition."""
import numpy as np

from ebwt2indel_amd import sim


def test_builder_matches_definition():
    rng = np.random.Generator(np.random.PCG64(3))
    reads = [sim.BASES[rng.integers(0, 4, size=int(rng.integers(1, 40)))].tobytes() for _ in range(200)]
    reads += [b"ACGT", b"ACGT", b"A", b"A", b"ACG"]      # duplicates and prefixes: tie-break by read index
    assert sim.build_ebwt(reads).tobytes() == sim.build_ebwt_naive(reads)


def test_builder_other_terminator():
    reads = [b"ACGTAC", b"ACGAAC", b"TTACG"]
    got = sim.build_ebwt(reads, term=36).tobytes()
    assert got == sim.build_ebwt_naive(reads).replace(b"#", b"$")


def test_simulator_is_deterministic():
    r1, g1, h1 = sim.diploid_sample(5000, 300, 50, seed=9)
    r2, g2, h2 = sim.diploid_sample(5000, 300, 50, seed=9)
    assert np.array_equal(r1, r2) and np.array_equal(g1, g2) and np.array_equal(h1, h2)
    assert not np.array_equal(g1[:len(h1)], h1[:len(g1)])
