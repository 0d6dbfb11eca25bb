"""The oracle's threaded index builder and traversal (checkers for full-size inputs, scripts/fullsize_check.py) against
its serial, reference-ordered code: same index bit for bit, same flag vectors, same counters."""
import numpy as np
import pytest

from ebwt2indel_amd import sim
from tests import _oracle
from tests.conftest import golden_bytes


@pytest.mark.parametrize("name,term", [("dip.ebwt", 35), ("edge.ebwt", 35), ("n128_t36.ebwt", 36), ("tiny.ebwt", 35), ("het_m.ebwt", 35)])
@pytest.mark.parametrize("threads", [2, 5])
def test_threaded_oracle_equals_serial(oracle, name, term, threads):
    e = golden_bytes(name)
    a, b = oracle.index(e, term), oracle.index(e, term, threads=threads)
    assert oracle.lib.orc_index_equal(a.h, b.h)
    fa = _oracle.OracleFlags(oracle, a, None, 16, 30)
    fb = _oracle.OracleFlags(oracle, b, None, 16, 30, threads=threads)
    for va, vb in zip(fa.export()[:3], fb.export()[:3]):
        assert np.array_equal(va, vb)
    ca, cb = fa.counters(), fb.counters()
    for k in ("leaves", "nodes", "lcp_values", "minima"):
        assert ca[k] == cb[k], k


def test_threaded_index_spans_several_chunks_and_reports_the_first_bad_symbol(oracle):
    # > 2^20 blocks = several builder chunks (2^27 symbols each); random symbols are enough for the index
    rng = np.random.Generator(np.random.PCG64(11))
    e = np.frombuffer(b"ACGT#", dtype=np.uint8)[rng.integers(0, 5, size=(1 << 28) + 12345)]
    a, b = oracle.index(e), oracle.index(e, threads=4)
    assert oracle.lib.orc_index_equal(a.h, b.h)
    del a, b
    e = e.copy()
    e[200_000_000] = ord("N")
    e[150_000_001] = ord("x")
    bad = oracle.index(e, threads=4)
    assert not bad.ok and bad.bad() == (ord("x"), 150_000_001)


def test_threaded_oracle_on_fresh_reads(oracle):
    reads, _, _ = sim.diploid_sample(60_000, 12_000, 100, seed=21)
    e = sim.build_ebwt(reads)
    a = oracle.index(e)
    fa = _oracle.OracleFlags(oracle, a, None, 16, 30)
    fb = _oracle.OracleFlags(oracle, a, None, 16, 30, threads=8)
    for va, vb in zip(fa.export()[:3], fb.export()[:3]):
        assert np.array_equal(va, vb)
    assert fa.counters()["nodes"] == fb.counters()["nodes"] and fa.counters()["minima"] == fb.counters()["minima"]


@pytest.mark.parametrize("a,b", [("hom_a.ebwt", "hom_b.ebwt"), ("het_a.ebwt", "het_b.ebwt"), ("edge_a.ebwt", "edge_b.ebwt")])
@pytest.mark.parametrize("threads", [2, 7])
def test_threaded_two_bwt_traversal_equals_serial(oracle, a, b, threads):
    x1, x2 = oracle.index(golden_bytes(a)), oracle.index(golden_bytes(b))
    fa = _oracle.OracleFlags(oracle, x1, x2, 16, 30)
    fb = _oracle.OracleFlags(oracle, x1, x2, 16, 30, threads=threads)
    for va, vb in zip(fa.export(), fb.export()):
        assert np.array_equal(va, vb)
    ca, cb = fa.counters(), fb.counters()
    for k in ("leaves", "nodes", "lcp_values", "da_values", "minima"):
        assert ca[k] == cb[k], k


@pytest.mark.parametrize("name", ["m1_default", "m1_edge_default", "m1_edge_m2", "m2_het_default", "m2_edge_default", "m3_het_default", "m3_hom_default"])
@pytest.mark.parametrize("shards", [3, 16])
def test_sharded_scan_renumbered_equals_the_reference_snp(oracle, name, shards):
    """scripts/fullsize_check.py's whole-range phase 4: the range cut into shards, each shard's clusters numbered from 1 (on
    any thread), the texts renumbered into the single pass's running cluster_nr -- byte for byte the reference-written .snp,
    in all three modes (mode 1 leaves gaps in the numbering, ebwt2InDel.cpp:1328; modes 2 / 3 do not, :1250)."""
    from concurrent.futures import ThreadPoolExecutor

    from ebwt2indel_amd import distributed as ebd
    from tests import _fullsize as fs
    from tests.conftest import CASES
    case = [c for c in CASES if c["name"] == name][0]
    p = _oracle.Params.from_args(case["args"])
    inp = case["inputs"]
    x1 = oracle.index(golden_bytes(inp["bwt1"]), p.term)
    x2 = oracle.index(golden_bytes(inp["bwt2"]), p.term) if case["mode"] == 2 else None
    da = golden_bytes(inp["da"]) if case["mode"] == 3 else None
    fl = _oracle.OracleFlags(oracle, x1, x2, p.K, p.k_right, da)
    n = x1.n + (x2.n if x2 is not None else 0)
    cuts = [ebd.shard_range(n, i, shards) for i in range(shards)]
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(lambda be: fl.scan(p, be[0], be[1]), cuts))
    text = fs.renumber([r["text"] for r in parts], [r["cluster_nr"] - 1 for r in parts])
    assert text == golden_bytes(case["snp"])


@pytest.mark.parametrize("n,piece", [(128 * 1000, 128 * 77), (128 * 1000 + 55, 128 * 300), (999, 128), (1 << 22, 1 << 20), ((1 << 25) * 3 + 17, (1 << 25) + 128)])
def test_streamed_index_equals_the_one_built_from_a_buffer(oracle, n, piece):
    """orc_index_stream_*: the index fed in pieces (how the 151 G-symbol eBWT of BASELINE configs[4] reaches the oracle: it comes off
    the GPU a few GB at a time and would not fit the host beside the index): bit-identical, whatever the order of the pieces, n a
    multiple of the block size or not; a bad symbol is reported with its position."""
    rng = np.random.Generator(np.random.PCG64(n))
    e = np.frombuffer(b"ACGT#", dtype=np.uint8)[rng.integers(0, 5, size=n)]
    want = oracle.index(e)
    x = _oracle.OracleIndex.stream(oracle, n)
    cuts = list(range(0, n, piece))
    for pos in reversed(cuts):
        x.append(pos, e[pos:pos + piece], threads=3)
    x.finish(threads=3)
    assert x.ok and oracle.lib.orc_index_equal(want.h, x.h)
    pos = rng.integers(0, n + 1, size=5000, dtype=np.uint64)
    assert np.array_equal(x.rank4(pos), want.rank4(pos))
    if n > 5000:
        e = e.copy()
        e[4321] = ord("N")
        y = _oracle.OracleIndex.stream(oracle, n)
        for pos in cuts:
            y.append(pos, e[pos:pos + piece], threads=2)
        y.finish(threads=2)
        assert not y.ok and y.bad() == (ord("N"), 4321)


@pytest.mark.parametrize("name", ["dip.ebwt", "edge.ebwt", "het_m.ebwt"])
def test_sampled_traversal_check_is_the_traversal(oracle, name):
    """orc_navigate_one_sample_check (the traversal check of BASELINE configs[4], where the whole traversal on the host is out of
    reach): with every subtree sampled it visits exactly the serial traversal's leaves, nodes, LCP slots and minima and finds the serial
    traversal's own vectors right; the subtrees of a partition of the frontier add up to the whole; and a flipped bit of any vector, in
    a leaf interval or on a node border, top of the tree or below the frontier, is found by the sample that owns it."""
    e = golden_bytes(name)
    x = oracle.index(e)
    f = _oracle.OracleFlags(oracle, x, None, 16, 30)
    vec = [v.copy() for v in f.export()[:3]]
    want = f.counters()
    whole = _oracle.traversal_sample_check(oracle, x, 16, 30, vec, every=1, frontier=64, threads=3)
    assert whole["mismatches"] == 0 and whole["first_mismatch_plus_1"] == 0
    assert (whole["leaves"], whole["nodes"], whole["lcp_slots"] + 1, whole["minima_expected"]) == (want["leaves"], want["nodes"], want["lcp_values"], want["minima"])
    top = _oracle.traversal_sample_check(oracle, x, 16, 30, vec, every=1 << 40, which=(1 << 40) - 1, frontier=64)        # no subtree at all: the top alone
    parts = [_oracle.traversal_sample_check(oracle, x, 16, 30, vec, every=5, which=w, frontier=64, threads=2) for w in range(5)]
    assert all(p_["mismatches"] == 0 for p_ in parts)
    for k in ("leaves", "nodes", "lcp_slots", "minima_expected"):
        assert sum(p_[k] - top[k] for p_ in parts) + top[k] == whole[k], k
    assert 0 < top["nodes"] < min(p_["nodes"] for p_ in parts)
    # flipped bits: every LCP slot 1 .. n-1 is some sample's, so the five samples together see each flip exactly once
    rng = np.random.Generator(np.random.PCG64(5))
    for which_vec in range(3):
        for pos in rng.integers(1, len(e), size=6):
            bad = [v.copy() for v in vec]
            bad[which_vec][pos >> 6] ^= np.uint64(1) << np.uint64(pos & 63)
            found = [_oracle.traversal_sample_check(oracle, x, 16, 30, bad, every=5, which=w, frontier=64) for w in range(5)]
            in_top = _oracle.traversal_sample_check(oracle, x, 16, 30, bad, every=1 << 40, which=(1 << 40) - 1, frontier=64)["mismatches"]
            hits = sum(r["mismatches"] for r in found)
            assert hits == (5 if in_top else 1), (which_vec, int(pos), hits, in_top)
            assert all(r["first_mismatch_plus_1"] in (0, ) or r["mismatches"] for r in found)
