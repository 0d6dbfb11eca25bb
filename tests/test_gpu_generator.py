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


# ---------------------------------------------------------------------------------------------------------------------------
# The slab-by-slab generator (tools/ebwt_gen_slabs.hip): what makes BASELINE configs[4] (1 G distinct reads, a text past 2^32
# symbols, 151 G keys) possible on one GPU.  Same outputs as the host builder for every choice of slab / group prefix length.

@pytest.mark.parametrize("genome_len,n_reads,read_len,seed,P,Pg", [
    (400, 60, 20, 1, -1, -1), (400, 60, 20, 1, 0, 0), (400, 60, 20, 1, 3, 1),
    (5000, 1500, 100, 2, 1, 0), (5000, 1500, 100, 2, 2, 2), (5000, 1500, 100, 2, 5, 3), (5000, 1500, 100, 2, 6, 0),
    (50_000, 10_000, 150, 3, -1, -1), (50_000, 10_000, 150, 3, 4, 2), (50_000, 10_000, 150, 3, 6, 4),
    (300_000, 120_000, 100, 4, 3, 1), (300_000, 120_000, 100, 4, 5, 3)])
def test_slabbed_generator_matches_host_builder(genome_len, n_reads, read_len, seed, P, Pg):
    text, starts = _case(genome_len, n_reads, read_len, seed, snp_every=200, indel_every=700)
    want = sim.build_ebwt(sim.reads_from_starts(text, starts, read_len))
    dev = sim.build_ebwt_gpu_slabbed(text, starts, read_len, prefix_len=P, group_len=Pg)
    got, _ = dev.to_host()
    dev.free()
    assert dev.n == len(want) == n_reads * (read_len + 1)
    if P >= 0:
        assert dev.info[:2] == (P, Pg)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("P,Pg", [(-1, -1), (2, 0), (4, 3)])
def test_slabbed_generator_repeats_and_document_array(P, Pg):
    """Exact repeats longer than a read, a run of 300 A (hundreds of text positions equal on every window: the tie sorter), equal
    reads (ties by read index) and the two-collection document array."""
    rng = np.random.Generator(np.random.PCG64(9))
    unit = sim.BASES[rng.integers(0, 4, size=700)]
    g = np.concatenate([unit, unit, sim.BASES[rng.integers(0, 4, size=300)], unit[:400], np.full(300, ord("A"), dtype=np.uint8)])
    h2 = g.copy()
    h2[900] = ord("A") if g[900] != ord("A") else ord("C")
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, 4000, 80, 5)
    reads = sim.reads_from_starts(text, starts, 80)
    want, want_da = sim.build_merged_ebwt(reads[:2500], reads[2500:])
    dev = sim.build_ebwt_gpu_slabbed(text, starts, 80, n_first=2500, want_da=True, prefix_len=P, group_len=Pg)
    got, da = dev.to_host()
    dev.free()
    assert np.array_equal(got, want)
    assert np.array_equal(da, want_da)


def test_slabbed_generator_refuses_a_text_that_repeats_itself():
    """More than 4096 text positions equal on 32 symbols cannot be put in order by comparison in reasonable time: refused, not hung."""
    g = np.full(6000, ord("A"), dtype=np.uint8)
    text, offs = sim.haplotype_text([g])
    starts = np.arange(0, 500, dtype=np.uint64)
    with pytest.raises(RuntimeError, match="repeats itself"):
        sim.build_ebwt_gpu_slabbed(text, starts, 50)


def test_slabbed_generator_equals_the_one_pass_generator():
    """2 M reads of 150 bp (n = 302 M): the two GPU generators agree byte for byte (the one-pass one is what every recorded
    full-size input was made with)."""
    text, starts = _case(10_000_000, 2_000_000, 150, 11)
    a = sim.build_ebwt_gpu(text, starts, 150)
    want, _ = a.to_host()
    a.free()
    b = sim.build_ebwt_gpu_slabbed(text, starts, 150, prefix_len=3, group_len=2)
    got, _ = b.to_host()
    b.free()
    assert np.array_equal(got, want)


def test_device_made_sample_and_slabbed_generator():
    """The sample made on the device (counter-based recipe): layout, the second haplotype's variants, reads inside their
    haplotypes -- and its eBWT through device pointers equals the host builder's of the same reads."""
    G, n_reads, L = 200_000, 60_000, 100
    s = sim.synth_diploid_gpu(G, n_reads, L, seed=7, snp_every=500, indel_every=2000)
    text, starts = s.text_to_host(), s.starts_to_host()
    assert len(text) == s.text_len == G + s.hap2_len + 2 and text[G] == 1 and text[-1] == 2
    h1, h2 = text[:G], text[G + 1:-1]
    assert set(np.unique(h1)) <= set(b"ACGT") and set(np.unique(h2)) <= set(b"ACGT")
    assert 0.23 < (h1 == ord("A")).mean() < 0.27
    assert abs(int(s.hap2_len) - G) <= 3 * (G // 2000 + 1)
    assert not np.array_equal(h1[-5000:], h2[-5000:])          # the indels shift the second haplotype
    # between two indels the haplotypes agree except at the substitutions: the first block up to its indel
    first_indel = 2000 // 2 + 500 // 4 + 7
    diff = np.nonzero(h1[:first_indel] != h2[:first_indel])[0]
    assert list(diff) == [250, 750]
    ev, od = starts[0::2], starts[1::2]
    assert ev.max() + L <= G and od.min() >= G + 1 and od.max() + L <= len(text) - 1
    want = sim.build_ebwt(sim.reads_from_starts(text, starts, L))
    dev = sim.build_ebwt_gpu_slabbed(s.text_ptr, s.starts_ptr, L, n_reads=n_reads, text_len=s.text_len, prefix_len=4, group_len=2)
    got, _ = dev.to_host()
    dev.free()
    s.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("genome_len,n_reads,read_len,seed,P,Pg", [(3000, 800, 60, 5, -1, -1), (3000, 800, 60, 5, 3, 1), (2000, 1500, 40, 6, 5, 2),
                                                              (60_000, 25_000, 100, 7, 2, 0), (60_000, 25_000, 100, 7, 6, 3)])
def test_slabbed_generator_states_the_lcp_vectors_of_the_oracle(genome_len, n_reads, read_len, seed, P, Pg):
    """The generator's own statement of thrK / thrR / minima -- common prefixes of adjacent read suffixes, from the suffix order of the
    TEXT -- equals what the CPU oracle (the reference's suffix-tree navigation of the eBWT) computes: bit for bit for the two
    threshold vectors, and for the minima everywhere but the few rows at the edges of slabs and short blocks the generator leaves
    undecided.  This is what lets it stand in as the checker of phases 2-3 at 151 G positions, where no oracle fits the host."""
    from tests import _oracle
    text, starts = _case(genome_len, n_reads, read_len, seed, snp_every=200, indel_every=700)
    dev = sim.build_ebwt_gpu_slabbed(text, starts, read_len, prefix_len=P, group_len=Pg, lcp_truth=(16, 30))
    ebwt, _ = dev.to_host()
    n, words = dev.n, (dev.n + 63) // 64
    got = []
    for ptr in dev.truth:
        a = np.empty(words + 1, dtype=np.uint64)
        sim._hip_copy_to_host(a, ptr, (words + 1) * 8)
        assert a[words] == 0
        got.append(a[:words])
    unknown = dev.unknown
    dev.free_truth()
    dev.free()
    orc = _oracle.Oracle()
    fl = _oracle.OracleFlags(orc, orc.index(ebwt), K=16, k_right=30)
    thrK, thrR, minima, _ = fl.export()
    assert np.array_equal(got[0], thrK[:words]) and np.array_equal(got[1], thrR[:words])
    assert len(unknown) and unknown.max() < n
    mask = np.full(words, ~np.uint64(0), dtype=np.uint64)
    np.bitwise_and.at(mask, (unknown >> np.uint64(6)).astype(np.int64), ~(np.uint64(1) << (unknown & np.uint64(63))))
    assert not (got[2] & ~mask).any()                      # undecided rows are left 0
    assert np.array_equal(got[2], minima[:words] & mask)
    assert int(np.unpackbits((minima[:words] & ~mask).view(np.uint8)).sum()) <= len(unknown)


@pytest.mark.parametrize("rc", [False, True])
@pytest.mark.parametrize("genome_len,n_reads,read_len,seed,prefix_len", [(5000, 1500, 100, 12, -1), (60_000, 20_000, 150, 13, 3), (60_000, 20_000, 150, 14, 5)])
def test_reads_with_errors_through_the_slabbed_generator(genome_len, n_reads, read_len, seed, prefix_len, rc):
    """Reads with 1 % substitution errors (and their reverse complements: the reference's documented input, README.md:88-100) are
    substrings of no genome: the generator's text is then the reads themselves, one behind the other.  The sample made on the GPU
    (sim.error_sample_gpu) is the host recipe's, byte for byte, and its eBWT the suffix-sorting host builder's."""
    text, starts = _case(genome_len, n_reads, read_len, seed, snp_every=200, indel_every=700)
    reads = sim.error_sample(text, starts, read_len, seed, 10_000, rc=rc)
    assert 0.005 < (reads[:n_reads] != sim.reads_from_starts(text, starts, read_len)).mean() < 0.02
    d_reads, d_starts = sim.error_sample_gpu(text, starts, read_len, seed, 10_000, rc=rc, chunk_reads=777)
    assert np.array_equal(d_reads.cpu().numpy()[:-1].reshape(-1, read_len), reads)
    want = sim.build_ebwt(reads)
    dev = sim.build_ebwt_gpu_slabbed(d_reads.data_ptr(), d_starts.data_ptr(), read_len, n_reads=len(reads), text_len=int(d_reads.numel()), prefix_len=prefix_len)
    got, _ = dev.to_host()
    dev.free()
    assert np.array_equal(got, want)
