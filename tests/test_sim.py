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


def test_replicating_every_read_repeats_every_bwt_symbol():
    """The identity behind sim.replicated_ebwt_gpu (inputs past 2^32 positions that the reference binary can still get
    through): with every read `c` times in a row, the eBWT is the base one with each symbol repeated c times."""
    reads, _, _ = sim.diploid_sample(3000, 400, 40, seed=4)
    base = sim.build_ebwt(reads)
    for c in (2, 3, 8):
        rep = np.repeat(reads, c, axis=0)
        assert np.array_equal(sim.build_ebwt(rep), np.repeat(base, c))
    ragged = [b"ACGT", b"ACG", b"ACGT", b"A", b"TTGACA", b"GACA"]     # shared suffixes, prefixes, duplicates in the base itself
    rep = [r for r in ragged for _ in range(3)]
    assert sim.build_ebwt(rep).tobytes() == np.repeat(np.frombuffer(sim.build_ebwt_naive(ragged), dtype=np.uint8), 3).tobytes()
