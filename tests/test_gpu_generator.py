"""Synthetic-input tooling on the GPU: the benchmark-scale eBWT generator (tools/ebwt_gen.hip) must
produce exactly the eBWT / DA of the same reads that the suffix-sorting host builder produces."""
import numpy as np
import pytest

from ebwt2indel_amd import sim

pytestmark = pytest.mark.gpu


def _case(genome_len, n_reads, read_len, seed, **kw):
    g = sim.random_genome(genome_len, seed)
    h2 = sim.mutate_haplotype(g, seed, **kw)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, n_reads, read_len, seed)
    return text, starts


@pytest.mark.parametrize("genome_len,n_reads,read_len,seed", [(400, 60, 20, 1), (5000, 1500, 100, 2), (50_000, 10_000, 150, 3),
                                                              (300_000, 120_000, 100, 4)])
def test_gpu_generator_matches_host_builder(genome_len, n_reads, read_len, seed):
    text, starts = _case(genome_len, n_reads, read_len, seed, snp_every=200, indel_every=700)
    want = sim.build_ebwt(sim.reads_from_starts(text, starts, read_len))
    dev = sim.build_ebwt_gpu(text, starts, read_len)
    got, _ = dev.to_host()
    dev.free()
    assert dev.n == len(want) == n_reads * (read_len + 1)
    assert np.array_equal(got, want)


def test_gpu_generator_repeats_and_document_array():
    """Exact repeats longer than a read (ties resolved by read index) and the two-collection document array."""
    rng = np.random.Generator(np.random.PCG64(9))
    unit = sim.BASES[rng.integers(0, 4, size=700)]
    g = np.concatenate([unit, unit, sim.BASES[rng.integers(0, 4, size=300)], unit[:400], np.full(300, ord("A"), dtype=np.uint8)])
    h2 = g.copy()
    h2[900] = ord("A") if g[900] != ord("A") else ord("C")
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, 4000, 80, 5)
    reads = sim.reads_from_starts(text, starts, 80)
    want, want_da = sim.build_merged_ebwt(reads[:2500], reads[2500:])
    dev = sim.build_ebwt_gpu(text, starts, 80, n_first=2500, want_da=True)
    got, da = dev.to_host()
    dev.free()
    assert np.array_equal(got, want)
    assert np.array_equal(da, want_da)


def test_gpu_generator_slabbed_sort(monkeypatch):
    """Force the bucket-by-lo + per-slab sort path used at benchmark scale."""
    monkeypatch.setenv("EBGEN_SLABS", "37")
    text, starts = _case(60_000, 20_000, 120, 6, snp_every=300, indel_every=900)
    want = sim.build_ebwt(sim.reads_from_starts(text, starts, 120))
    dev = sim.build_ebwt_gpu(text, starts, 120)
    got, _ = dev.to_host()
    dev.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("slabs,rbits", [("64", None), ("1000", "40"), ("3", "47")])
def test_gpu_generator_slab_relative_keys(monkeypatch, slabs, rbits):
    """The key layout of the large configurations forced on a small input: many slabs with only lo's offset inside the slab
    in the key, and a read-index field as wide as 100 M / 1 G reads need (EBGEN_RBITS; with 47 bits the generator must
    itself raise the number of slabs until (lo in slab, length, read index) fits 64 bits)."""
    monkeypatch.setenv("EBGEN_SLABS", slabs)
    if rbits:
        monkeypatch.setenv("EBGEN_RBITS", rbits)
    text, starts = _case(80_000, 30_000, 110, 8, snp_every=300, indel_every=900)
    reads = sim.reads_from_starts(text, starts, 110)
    want, want_da = sim.build_merged_ebwt(reads[:13_000], reads[13_000:])
    dev = sim.build_ebwt_gpu(text, starts, 110, n_first=13_000, want_da=True)
    got, da = dev.to_host()
    dev.free()
    assert np.array_equal(got, want)
    assert np.array_equal(da, want_da)
