"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of
libebwt2indel.so; the CPU oracle (pinned to the reference by tests/test_oracle_golden.py) is the checker.
Bit-exact everywhere: this path is integer / byte work."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from ebwt2indel_amd import _lib, api, sim
from tests import _oracle
from tests.conftest import CASES, CASE_IDS, golden_bytes

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _positions(n, m, seed, upto_n=True):
    r = _rng(seed)
    hi = n + 1 if upto_n else n
    pos = r.integers(0, hi, size=m, dtype=np.uint64)
    edge = np.array([0, 1, 31, 32, 63, 64, 127, 128, 129, n - 1 if n else 0, n if upto_n else max(n - 1, 0)], dtype=np.uint64)
    return np.concatenate([edge[edge < hi], pos])


@pytest.fixture(scope="module")
def dip(oracle):
    e = golden_bytes("dip.ebwt")
    return e, api.DnaBwt(e), oracle.index(e)


# ------------------------------------------------------------------ phase 1 + primitives

@pytest.mark.parametrize("name,term", [("dip.ebwt", 35), ("edge.ebwt", 35), ("n128_t36.ebwt", 36), ("tiny.ebwt", 35)])
def test_index_F_and_size(oracle, name, term):
    e = golden_bytes(name)
    g, o = api.DnaBwt(e, term=term), oracle.index(e, term)
    assert g.size() == len(e) == o.n
    assert g.F_arr == [int(x) for x in o.F()]
    assert g.n_blocks == (len(e) + 128) // 128


def test_index_from_file_and_device_buffer(tmp_path, dip):
    e, g, o = dip
    path = tmp_path / "x.ebwt"
    path.write_bytes(e)
    gf = api.DnaBwt(str(path))
    pos = _positions(len(e), 5000, 1)
    assert np.array_equal(gf.parallel_rank(pos), o.rank4(pos))
    torch = pytest.importorskip("torch")
    t = torch.frombuffer(bytearray(e), dtype=torch.uint8).cuda()
    gd = api.DnaBwt(None, device_ptr=t.data_ptr(), n=t.numel())
    assert gd.F_arr == g.F_arr
    assert np.array_equal(gd.parallel_rank(pos), o.rank4(pos))


def test_threaded_pinned_upload_gives_the_same_index(monkeypatch, dip):
    """Large inputs go to the GPU through page-locked pieces filled by host threads; forced here on a small one, with
    pieces that do not divide it."""
    e, g, _ = dip
    monkeypatch.setenv("EB_INDEX_PINNED", f"1,{128 * 37},3")       # from 1 byte on, pieces of 4736 symbols, 3 reader threads
    g2 = api.DnaBwt(e)
    assert g2.F_arr == g.F_arr and g2.n == g.n
    pos = _positions(len(e), 20_000, 5)
    assert np.array_equal(g2.parallel_rank(pos), g.parallel_rank(pos))
    idx = _positions(len(e), 5_000, 6, upto_n=False)
    assert np.array_equal(g2[idx], g[idx])


def test_forbidden_character_is_reported():
    e = bytearray(golden_bytes("dip.ebwt"))
    e[70001] = ord("N")
    e[90000] = ord("x")
    with pytest.raises(_lib.EbError) as err:
        api.DnaBwt(bytes(e))
    assert err.value.code == _lib.EB_ERR_ALPHABET
    assert api.DnaBwt.bad_symbol() == (ord("N"), 70001)
    assert "forbidden character 'N' (ASCII code 78)" in str(err.value)


def test_parallel_rank_exhaustive_small(oracle):
    e = golden_bytes("tiny.ebwt") + golden_bytes("edge.ebwt")[:1000]
    g, o = api.DnaBwt(e), oracle.index(e)
    pos = np.arange(len(e) + 1, dtype=np.uint64)
    got = g.parallel_rank(pos)
    assert np.array_equal(got, o.rank4(pos))
    naive = np.zeros((len(e) + 1, 4), dtype=np.uint64)        # definition: prefix counts
    arr = np.frombuffer(e, dtype=np.uint8)
    for c, ch in enumerate(b"ACGT"):
        naive[1:, c] = np.cumsum(arr == ch)
    assert np.array_equal(got, naive)


def test_parallel_rank_random(dip):
    e, g, o = dip
    pos = _positions(len(e), 200_000, 2)
    assert np.array_equal(g.parallel_rank(pos), o.rank4(pos))


def test_access(dip):
    e, g, o = dip
    pos = _positions(len(e), 50_000, 3, upto_n=False)
    got = g[pos]
    assert np.array_equal(got, np.frombuffer(e, dtype=np.uint8)[pos.astype(np.int64)])
    assert np.array_equal(got, o.access(pos))


def test_lf_range_and_sym(dip):
    e, g, o = dip
    r = _rng(4)
    l = r.integers(0, len(e) + 1, size=60_000, dtype=np.uint64)
    w = r.integers(0, 300, size=60_000, dtype=np.uint64)
    w[:2000] = 0                                               # empty ranges take the one-rank path
    rr = np.minimum(l + w, np.uint64(len(e)))
    lr = np.stack([l, rr], axis=1)
    assert np.array_equal(g.LF(lr), o.lf_range(lr))
    sym = np.frombuffer(b"ACGT", dtype=np.uint8)[r.integers(0, 4, size=len(lr))]
    assert np.array_equal(g.LF(lr, sym), o.lf_range_sym(lr, sym))


def test_fl_is_inverse_of_lf(dip):
    e, g, o = dip
    r = _rng(5)
    pos = r.integers(g.F_arr[0], len(e), size=40_000, dtype=np.uint64)
    fl = g.FL(pos)
    assert np.array_equal(fl, o.fl(pos))
    # LF(FL(i)) == i : row fl holds the letter F(i); its LF image is i
    arr = np.frombuffer(e, dtype=np.uint8)
    letters = arr[fl.astype(np.int64)]
    back = g.LF(np.stack([fl, fl + np.uint64(1)], axis=1), letters)[:, 0]
    assert np.array_equal(back, pos)


@pytest.mark.parametrize("shift", ["5", "7", "10"])
@pytest.mark.parametrize("name", ["dip.ebwt", "edge.ebwt", "het_m.ebwt"])
def test_fl_under_every_select_sampling(oracle, monkeypatch, name, shift):
    """The select directory samples every 32nd occurrence of a letter up to 2^35 symbols and every 128th beyond
    (EB_SELECT overrides): FL, and a whole run's .snp through the right-context walk, must not depend on it."""
    monkeypatch.setenv("EB_SELECT", shift)
    e = golden_bytes(name)
    g, o = api.DnaBwt(e), oracle.index(e)
    pos = _rng(int(shift)).integers(g.F_arr[0], len(e), size=30_000, dtype=np.uint64)
    assert np.array_equal(g.FL(pos), o.fl(pos))
    if name == "dip.ebwt":
        want, _ = oracle.run(e)
        assert api.run_one_dataset(e) == want


@pytest.mark.parametrize("shift,bits", [("5", "29"), ("7", "29"), ("7", "27"), ("5", "32"), ("10", "26")])
def test_fl_under_every_directory_entry_layout(oracle, monkeypatch, shift, bits):
    """A select-directory entry keeps the block of its sample in its low bits and the sample's ordinal inside that block in
    the bits an index of that size leaves free, in units of 1 (up to 2^32 symbols) to 16 (up to 2^36), none beyond
    (eb_select_guess); EB_SELECT=<shift>,<bits> puts the layouts of the large indexes on a small one.  FL, and a whole run's .snp
    through the right-context walk, must not depend on the layout: the guess only chooses where the probes start."""
    monkeypatch.setenv("EB_SELECT", f"{shift},{bits}")
    e = golden_bytes("dip.ebwt")
    g, o = api.DnaBwt(e), oracle.index(e)
    pos = _rng(int(shift) * 37 + int(bits)).integers(g.F_arr[0], len(e), size=30_000, dtype=np.uint64)
    assert np.array_equal(g.FL(pos), o.fl(pos))
    want, _ = oracle.run(e)
    assert api.run_one_dataset(e) == want


def _gappy_bwt(seed, n=300_000):
    """A legal input whose letters come in long stretches: many occurrences lie 64 rows or more behind their predecessor
    (void chunks of the FL table), others a row apart, with terminators sprinkled in."""
    r = _rng(seed)
    out = np.empty(n, dtype=np.uint8)
    i = 0
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    while i < n:
        kind = r.integers(0, 3)
        ln = int(r.integers(1, 400)) if kind == 0 else int(r.integers(1, 70))
        ln = min(ln, n - i)
        if kind == 2:
            out[i:i + ln] = letters[r.integers(0, 4, size=ln)]
        else:
            out[i:i + ln] = letters[r.integers(0, 4)]
        i += ln
    # a letter that shows up once every 600 rows (fifteen long jumps in one chunk), and one that stays away for 40 000 rows
    out[100_000:130_000] = ord("C")
    out[100_000:130_000:600] = ord("A")
    out[150_000:190_000][out[150_000:190_000] == ord("G")] = ord("T")
    out[r.integers(0, n, size=n // 150)] = 35
    return out.tobytes()


@pytest.mark.parametrize("name", ["dip.ebwt", "edge.ebwt", "het_m.ebwt", "n128_t36.ebwt", "tiny.ebwt", "gappy"])
def test_fl_through_the_fl_table(oracle, name):
    """FL with the table of repeatedly scanned indexes (eb_index_build_psi: one 16-byte chunk per 16 occurrences of a letter) equals
    FL through the select directory and the oracle's bisection (dna_string.hpp:254-272) -- at EVERY non-terminator row, so every
    chunk, every bitmap, every jump field and every void chunk (jumps that do not fit: the directory answers) is read."""
    term = 36 if name == "n128_t36.ebwt" else 35
    e = _gappy_bwt(11) if name == "gappy" else golden_bytes(name)
    g, o = api.DnaBwt(e, term=term), oracle.index(e, term)
    rows = np.arange(g.F_arr[0], len(e), dtype=np.uint64)
    plain = g.FL(rows)
    assert g.psi() == (False, 0, 0)
    g.build_psi()
    have, nbytes, voids = g.psi()
    assert have and nbytes >= len(e) - g.F_arr[0]
    assert voids >= 2 if name == "gappy" else voids * 50 * 16 < len(e)      # (read data: ~0.1 % of the chunks)
    assert np.array_equal(g.FL(rows), plain)
    some = rows[:: max(1, len(rows) // 50_000)]
    assert np.array_equal(plain[:: max(1, len(rows) // 50_000)], o.fl(some))
    g.build_psi()                                                     # idempotent
    assert g.psi() == (have, nbytes, voids)


def test_third_job_on_an_index_builds_the_fl_table(monkeypatch, dip):
    """A caller that keeps scanning an index gets the FL table with its third job (a one-shot run never pays for it); the call
    set does not depend on it, the right-context walk then issues about one memory request per FL step instead of two."""
    monkeypatch.delenv("EB_PSI", raising=False)
    e, _, _ = dip
    g = api.DnaBwt(e)
    p = api.make_params()
    fl = api.navigate_one_bwt(g, p)
    blobs, reqs = [], []
    for k in range(4):
        assert g.psi()[0] == (k >= 3)
        calls = api.scan_clusters(g, fl, p)
        blobs.append(bytes(calls.blob()))
        reqs.append(calls.stats.requests_right)
    assert blobs[1:] == blobs[:1] * 3
    assert reqs[0] == reqs[1] and reqs[2] == reqs[3] and reqs[2] < 0.6 * reqs[0]
    monkeypatch.setenv("EB_PSI", "0")
    g2 = api.DnaBwt(e)
    for k in range(4):
        assert bytes(api.scan_clusters(g2, fl, p).blob()) == blobs[0]
    assert not g2.psi()[0]


def test_lf_node(dip):
    e, g, o = dip
    r = _rng(6)
    m = 20_000
    b = np.sort(r.integers(0, len(e) + 1, size=(m, 6), dtype=np.uint64), axis=1)
    b[: m // 4, 1] = b[: m // 4, 0]                            # repeated boundaries share a rank query
    b[: m // 8, 5] = b[: m // 8, 4]
    nodes = np.concatenate([b, r.integers(0, 100, size=(m, 1), dtype=np.uint64)], axis=1)
    nodes[0] = g.root()
    assert np.array_equal(g.LF_node(nodes), o.lf_node(nodes))


# ------------------------------------------------------------------ phases 2-3

def _flag_case(oracle, case):
    p = _oracle.Params.from_args(case["args"])
    inp = case["inputs"]
    e1 = golden_bytes(inp["bwt1"])
    e2 = golden_bytes(inp["bwt2"]) if case["mode"] == 2 else None
    da = golden_bytes(inp["da"]) if case["mode"] == 3 else None
    return p, e1, e2, da


FLAG_CASES = [c for c in CASES if c["name"] in ("m1_default", "m1_k20", "m1_L40_R35", "m1_edge_default", "m1_n128_term36",
                                                "m2_hom_default", "m2_het_default", "m2_edge_default", "m2_hom_L25_R20_k14",
                                                "m3_het_default", "m1_tiny_nocluster")]


NAV_MODES = {
    # default: writers BELOW the thresholds mark their positions in interleaved bit words (atomicOr), inverted at the end;
    # leaves and nodes side by side on two streams
    "complement": {},
    # the reference's form (also the fallback for inputs that are no eBWT): writers at or above the thresholds set their bits --
    # one byte per position with plain stores (needs n spare bytes of HBM), or bit words with atomicOr when memory is short
    "classic_bytes": {"EB_NAV_FORM": "bytes"},
    "classic_words": {"EB_NAV_FORM": "words"},
    # the round-3 one-BWT node kernel (every node through the stack) instead of the chain-following one of round 4; the
    # chain-following walk thrown away as if a launch had overflowed its stack (the caller then walks again the round-3 way); and
    # the chain-following walk REALLY overflowing (told it has room for 64 records per launch: records dropped in mid-launch, the
    # error raised on the device, the group queued ahead idling, the leaf thread joined) before the round-3 kernels walk again
    "round3_node_kernel": {"EB_NAV_CHAINS": "0"},
    "chains_overflow_walks_again": {"EB_NAV_CHAINS": "2"},
    "chains_really_overflow": {"EB_NAV_CHAINS": "3"},
    "chains_really_overflow_classic_words": {"EB_NAV_CHAINS": "3", "EB_NAV_FORM": "words"},
    "classic_words_round3_kernels": {"EB_NAV_FORM": "words", "EB_NAV_CHAINS": "0"},
}


@pytest.mark.parametrize("nav_mode", list(NAV_MODES))
@pytest.mark.parametrize("case", FLAG_CASES, ids=lambda c: c["name"])
def test_flag_vectors_bit_exact(oracle, monkeypatch, case, nav_mode):
    """Every way the traversal collects the LCP thresholds gives the oracle's vectors, bit for bit."""
    for k, v in NAV_MODES[nav_mode].items():
        monkeypatch.setenv(k, v)
    p, e1, e2, da = _flag_case(oracle, case)
    o1 = oracle.index(e1, p.term)
    o2 = oracle.index(e2, p.term) if e2 is not None else None
    of = _oracle.OracleFlags(oracle, o1, o2, p.K, p.k_right, da)
    g1 = api.DnaBwt(e1, term=p.term)
    g2 = api.DnaBwt(e2, term=p.term) if e2 is not None else None
    gf = api.Flags.build(g1, g2, p.K, p.k_right)
    if da is not None:
        gf.set_da(da)
    want = of.export()
    got = gf.export()
    for name, a, b in zip(("thrK", "thrR", "minima", "da"), got, want):
        assert (a is None) == (b is None), name
        if a is not None:
            assert np.array_equal(a, b), name
    oc, st = of.counters(), gf.stats
    n = len(e1) + (len(e2) if e2 is not None else 0)
    assert (st.leaves, st.nodes, st.minima) == (oc["leaves"], oc["nodes"], oc["minima"])
    assert st.lcp_values == oc["lcp_values"] == n              # every LCP slot written exactly once
    if e2 is not None:
        assert st.da_values == oc["da_values"] == n


def test_flags_batch_size_does_not_change_result(oracle, monkeypatch):
    e = golden_bytes("dip.ebwt")
    g = api.DnaBwt(e)
    ref = api.Flags.build(g, None, 16, 30).export()
    for batch, mode in (("64", "complement"), ("1000", "classic_bytes"), ("777", "classic_words"), ("300,3", "complement"), ("200", "classic_words_round3_kernels"),
                        ("130", "chains_really_overflow"), ("64", "round3_node_kernel"), ("17", "complement")):
        monkeypatch.setenv("EB_NAV_BATCH", batch)                 # tiny batches: many launches, depth-first order ("300,3": and a three-batch arena)
        for k in ("EB_NAV_FORM", "EB_NAV_CHAINS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in NAV_MODES[mode].items():
            monkeypatch.setenv(k, v)
        got = api.Flags.build(g, None, 16, 30).export()
        for a, b in zip(got[:3], ref[:3]):
            assert np.array_equal(a, b)


NOT_EBWT = {
    # legal input files (the reference takes any string over A,C,G,T,TERM) that are the eBWT of no read collection and whose
    # traversal leaves LCP slots unwritten (oracle: lcp_values 1723 / 4999 / 1 of n)
    "runs": lambda r: np.frombuffer((b"A" * 50 + b"C" * 50 + b"#" * 3 + b"G" * 50 + b"T" * 50) * 10, dtype=np.uint8),
    "hardly_any_terminator": lambda r: r.choice(np.frombuffer(b"ACGT#", dtype=np.uint8), size=5000, p=[0.2499] * 4 + [0.0004]),
    "one_letter": lambda r: np.frombuffer(b"A" * 300, dtype=np.uint8),
}


@pytest.mark.parametrize("name", list(NOT_EBWT))
def test_input_that_is_no_ebwt_falls_back_to_the_reference_form(oracle, monkeypatch, name):
    """When the traversal does not write every LCP slot the reference leaves the others at 0; the complement form (which
    would leave them at 1) must notice (lcp_values != n) and walk again the reference's way."""
    for seed in range(40):                                        # the random one: first seed that leaves slots unwritten
        e = NOT_EBWT[name](np.random.Generator(np.random.PCG64(seed)))
        o = oracle.index(e)
        of = _oracle.OracleFlags(oracle, o, None, 16, 30)
        oc = of.counters()
        if oc["lcp_values"] != len(e):
            break
    assert oc["lcp_values"] != len(e)                              # the premise
    g = api.DnaBwt(e)
    got = api.Flags.build(g, None, 16, 30)
    monkeypatch.setenv("EB_NAV_FORM", "bytes")
    classic = api.Flags.build(g, None, 16, 30)
    for a, b, c in zip(got.export()[:3], classic.export()[:3], of.export()[:3]):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    assert got.stats.lcp_values == classic.stats.lcp_values == oc["lcp_values"]
    want, _ = oracle.run(e)
    monkeypatch.delenv("EB_NAV_FORM")
    assert api.run_one_dataset(e) == want


def test_two_files_that_are_no_ebwt_pair_are_refused_before_phase_4():
    """Two-BWT mode addresses the BWTs by document-array ranks; two legal files that are not the eBWTs of two collections can
    leave the rebuilt array short of ones (the reference would then read past its vectors).  The library checks the count
    and refuses phase 4 with a message -- it never lets a kernel index a BWT out of range."""
    refused = 0
    for name, make in NOT_EBWT.items():
        for seed in (0, 1, 2):
            e1 = make(np.random.Generator(np.random.PCG64(seed)))
            e2 = make(np.random.Generator(np.random.PCG64(100 + seed)))[: len(e1) - 37]
            g1, g2 = api.DnaBwt(e1), api.DnaBwt(e2)
            p = api.make_params()
            fl = api.navigate_two_bwts(g1, g2, p)
            ones = fl.popcounts()[3]
            if ones == g2.n:
                api.scan_clusters(g1, fl, p, bwt2=g2)               # consistent: phase 4 runs
            else:
                refused += 1
                with pytest.raises(api.EbError, match="document array"):
                    api.scan_clusters(g1, fl, p, bwt2=g2)
    assert refused > 0                                              # the premise: some pair does leave the array short


def test_short_document_array_repeats_its_last_byte(oracle):
    """A -d file shorter than the BWT: the reference's failed read() leaves its one-byte buffer unchanged, so the rest of
    the DA repeats the last byte read (ebwt2InDel.cpp:1503-1508; oracle pinned on this against the reference binary in
    tests/test_oracle_vs_ref.py)."""
    m, da = golden_bytes("het_m.ebwt"), golden_bytes("het_m.da")
    for cut, last in ((len(da) * 2 // 3, b"1"), (len(da) * 2 // 3 + 5, b"0"), (len(da) // 2 + 64, b"1"), (1, b"1")):
        d = bytearray(da[:cut])
        d[-1] = last[0]
        want, _ = oracle.run(m, da=bytes(d))
        assert api.run_two_datasets_da(m, bytes(d)) == want
        g = api.DnaBwt(m)
        fl = api.Flags.build(g, None, 16, 30)
        fl.set_da(bytes(d))
        bits = np.unpackbits(fl.export()[3].view(np.uint8), bitorder="little")[: len(m)]
        assert bits[cut:].all() == (last == b"1") and bits[cut:].any() == (last == b"1")
        assert np.array_equal(bits[:cut], (np.frombuffer(bytes(d), dtype=np.uint8) == ord("1")).astype(np.uint8))


@pytest.mark.parametrize("pieces", ["plain", "page_locked_pieces"])
def test_document_array_from_buffer_file_and_device_is_the_same_array(monkeypatch, tmp_path, pieces):
    """The -d array reaches the flags from a host buffer, from the file itself (page-locked pieces filled by the loader's reader
    threads: what the executable does) or from ASCII bytes already in HBM: one packed array, short and empty sources included
    (ebwt2InDel.cpp:1497-1508)."""
    import torch
    if pieces == "page_locked_pieces":
        monkeypatch.setenv("EB_INDEX_PINNED", f"1,{64 * 53},3")     # from 1 byte on, pieces of 3392 positions, 3 reader threads
    m, da = golden_bytes("het_m.ebwt"), golden_bytes("het_m.da")
    g = api.DnaBwt(m)
    fl = api.Flags.build(g, None, 16, 30)
    junk = bytearray(da)
    junk[100:110] = b"x2\n 1-1\x00\xff1"                           # anything but '1' is 0
    for src in (da, bytes(junk), da[: len(da) * 2 // 3 - 1] + b"1", da[: len(da) // 2 + 17] + b"0", da[:1], b"1", b"", da + b"1111"):
        lim = min(len(src), len(m))
        want = np.zeros(len(m), dtype=np.uint8)
        want[:lim] = np.frombuffer(src[:lim], dtype=np.uint8) == ord("1")
        if 0 < lim < len(m):
            want[lim:] = src[lim - 1] == ord("1")
        got = []
        fl.set_da(src)
        got.append(fl.export()[3].copy())
        pth = tmp_path / "x.da"
        pth.write_bytes(src)
        fl.set_da(b"0" * len(m))                                    # (each form starts from another array)
        fl.set_da_file(str(pth))
        got.append(fl.export()[3].copy())
        fl.set_da(b"1" * len(m))
        t = torch.frombuffer(bytearray(src) if src else bytearray(b"-"), dtype=torch.uint8).cuda()
        fl.set_da_device(t.data_ptr(), len(src))
        got.append(fl.export()[3].copy())
        # staged ahead (on another thread, as the executable does beside its traversal), then handed over
        import threading
        box = []
        th = threading.Thread(target=lambda: box.append(api.StagedDa(str(pth), len(m))))
        th.start()
        th.join()
        fl.set_da(b"0" * len(m))
        fl.set_da_staged(box[0])
        got.append(fl.export()[3].copy())
        for words in got:
            assert np.array_equal(np.unpackbits(words.view(np.uint8), bitorder="little")[: len(m)], want)
            assert not np.unpackbits(words.view(np.uint8), bitorder="little")[len(m):].any()
    with pytest.raises(api.EbError, match="could not read file"):
        fl.set_da_file(str(tmp_path / "absent.da"))
    with pytest.raises(api.EbError, match="could not read file"):
        api.StagedDa(str(tmp_path / "absent.da"), len(m))
    with pytest.raises(api.EbError, match="16-byte aligned"):
        fl.set_da_device(t.data_ptr() + 1, 100)
    wrong = api.StagedDa(str(tmp_path / "x.da"), len(m) + 64)
    with pytest.raises(api.EbError, match="staged document array was made for"):
        fl.set_da_staged(wrong)
    # and the call set that comes out of the file form is the reference's
    fl.set_da_file(str(tmp_path / "x.da"))                          # (da + b"1111": longer than the BWT, the excess is ignored)
    p = api.make_params()
    assert api.scan_clusters(g, fl, p).format_snp(p, True)[0] == api.run_two_datasets_da(m, da)


def test_flags_import_export_round_trip(oracle):
    e = golden_bytes("het_m.ebwt")
    da = golden_bytes("het_m.da")
    of = _oracle.OracleFlags(oracle, oracle.index(e), None, 16, 30, da)
    want = of.export()
    f = api.Flags.from_host(len(e), *want)
    for a, b in zip(f.export(), want):
        assert np.array_equal(a, b)


# ------------------------------------------------------------------ phase 4 and the whole path

@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_snp_byte_identical_to_reference(case):
    """End to end on the GPU (index -> flags -> scan -> writer) against the reference-written golden .snp."""
    p = api.make_params(case["args"])
    inp = case["inputs"]
    if case["mode"] == 1:
        got = api.run_one_dataset(golden_bytes(inp["bwt1"]), p)
    elif case["mode"] == 2:
        got = api.run_two_datasets(golden_bytes(inp["bwt1"]), golden_bytes(inp["bwt2"]), p)
    else:
        got = api.run_two_datasets_da(golden_bytes(inp["bwt1"]), golden_bytes(inp["da"]), p)
    assert got == golden_bytes(case["snp"])


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m1_edge_m2", "m2_het_default", "m2_edge_default",
                                                                      "m3_het_default", "m3_edge_default")], ids=lambda c: c["name"])
@pytest.mark.parametrize("gated,psi", [(False, "0"), (True, "0"), (False, "1"), (True, "1")],
                         ids=["walks_every_cluster", "walks_gated", "fl_table_every_cluster", "fl_table_gated"])
def test_cluster_records_match_oracle(oracle, monkeypatch, case, gated, psi):
    """Record-level parity of phase 4 (clusters, supports, contexts, counters), whole range and sharded -- with the right-context
    walk for every cluster (small call sets) and for the clusters that can print only (large call sets; forced on these small
    ones by EB_CONTEXT=gate=0: `gated`) -- each through the directory select (EB_PSI=0) and through the FL table of
    repeatedly scanned indexes (round 4, EB_PSI=1: built with the first job instead of the third)."""
    monkeypatch.setenv("EB_PSI", psi)
    if gated:
        monkeypatch.setenv("EB_CONTEXT", "gate=0")
    p, e1, e2, da = _flag_case(oracle, case)
    pp = api.make_params(case["args"])
    o1 = oracle.index(e1, p.term)
    o2 = oracle.index(e2, p.term) if e2 is not None else None
    of = _oracle.OracleFlags(oracle, o1, o2, p.K, p.k_right, da)
    g1 = api.DnaBwt(e1, term=p.term)
    g2 = api.DnaBwt(e2, term=p.term) if e2 is not None else None
    gf = api.Flags.from_host(of.words * 0 + len(e1) + (len(e2) if e2 is not None else 0), *of.export())   # oracle flags: isolates phase 4
    n = g1.n + (g2.n if g2 is not None else 0)
    skipped = 0
    for cuts in ([0, n], [0, n // 3, n // 2 + 17, n]):
        text, cluster_nr, events = b"", 1, 0
        plain = []
        for b, e in zip(cuts[:-1], cuts[1:]):
            want = of.scan(p, b, e, capture=True)
            if gated:
                # the path of large call sets: the walk runs behind the left-context extension and skips the clusters that
                # cannot print (EB_RIGHT_SKIPPED); the golden .snp below shows that nothing the writer needs is missing
                want = _oracle.apply_right_gate(want, p.k_left, p.k_right, case["mode"])
                skipped += want["skipped"]
            calls = api.scan_clusters(g1, gf, pp, bwt2=g2, begin=b, end=e)
            cl, chains, left, right = calls.arrays()
            assert len(calls) == want["count"]
            assert np.array_equal(cl, want["clusters"])
            assert np.array_equal(chains, want["chains"])
            assert np.array_equal(left, want["left"])
            assert np.array_equal(right, want["right"])
            assert bytes(calls.blob()) == _oracle.pack_calls(want, p.k_left, p.k_right)   # wire image, byte for byte
            st = calls.stats
            assert dict(runs_closed=st.runs_closed, clusters=st.clusters, bases_in_runs=st.bases_in_runs,
                        survivors=st.survivors, chains=st.chains) == want["stats"]
            t, cluster_nr, events = calls.format_snp(pp, case["mode"] != 1, cluster_nr, events)
            text += t
            plain.append(bytes(calls.blob()))
            if gated and len(calls):                   # the compact wire image carries the flag too
                job = api.scan_clusters_begin(g1, gf, pp, bwt2=g2, begin=b, end=e, device_only=True, wire=True)
                job.wait()
                ptr, length = job.device_wire()
                wire = np.empty(length, dtype=np.uint8)
                sim._hip_copy_to_host(wire, ptr, length)
                job.finish()
                assert bytes(api.Calls.unpack(wire.tobytes()).blob()) == bytes(calls.blob())
        assert text == golden_bytes(case["snp"])
        # the same shards as jobs of a pipelining caller (EB_SCAN_PIPELINED), all begun before the first is ended
        jobs = [api.scan_clusters_begin(g1, gf, pp, bwt2=g2, begin=b, end=e, pipelined=True) for b, e in zip(cuts[:-1], cuts[1:])]
        assert [bytes(j.finish().blob()) for j in jobs] == plain
    assert g1.psi()[0] == (psi == "1") and (g2 is None or g2.psi()[0] == (psi == "1"))
    if gated and case["name"] in ("m1_default", "m2_het_default", "m3_het_default"):
        assert skipped > 0                                           # the gate is exercised, not just present


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m1_edge_m2", "m2_het_default", "m3_het_default", "m3_edge_default")], ids=lambda c: c["name"])
def test_extension_by_one_lane_per_chain_gives_the_same_call_set(monkeypatch, case):
    """k_extend1 (one lane per left-context chain, the whole 64-byte block per lane) against k_extend (a quad per chain): the
    same call-set image byte for byte, the same number of block requests, and the reference-written .snp."""
    p = api.make_params(case["args"])
    inp = case["inputs"]
    g1 = api.DnaBwt(golden_bytes(inp["bwt1"]), term=p.term)
    g2 = api.DnaBwt(golden_bytes(inp["bwt2"]), term=p.term) if case["mode"] == 2 else None
    fl = api.navigate_two_bwts(g1, g2, p) if g2 is not None else api.navigate_one_bwt(g1, p)
    if case["mode"] == 3:
        fl.set_da(golden_bytes(inp["da"]))
    out = {}
    for lanes in ("4", "1"):
        monkeypatch.setenv("EB_CONTEXT", f"extend={lanes}")
        calls = api.scan_clusters(g1, fl, p, bwt2=g2)
        out[lanes] = (bytes(calls.blob()), calls.stats.requests_extend, calls.format_snp(p, case["mode"] != 1)[0])
    assert out["1"] == out["4"]
    assert out["1"][2] == golden_bytes(case["snp"])


def test_cluster_histogram_matches_oracle_counts(oracle):
    e = golden_bytes("dip.ebwt")
    g = api.DnaBwt(e)
    p = api.make_params()
    calls = api.scan_clusters(g, api.Flags.build(g, None, p.K, p.k_right), p)
    st = calls.stats
    hist = np.array(list(st.hist), dtype=np.uint64)
    assert hist[0] == 0 and all(hist[k] % k == 0 for k in range(1, 201))   # CLUST_SIZES[len] += len
    assert hist.sum() <= st.bases_in_runs


def test_cli_binary_byte_identical(tmp_path):
    """The ebwt2InDel executable (same flags as the reference) on the three modes."""
    def run(args, out):
        r = subprocess.run([_lib.CLI_PATH] + args + ["-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        return out.read_bytes(), r.stdout
    for name in ("dip.ebwt", "hom_a.ebwt", "hom_b.ebwt", "hom_m.ebwt", "hom_m.da", "tiny.ebwt"):
        (tmp_path / name).write_bytes(golden_bytes(name))
    got, log = run(["-1", str(tmp_path / "dip.ebwt"), "-m", "5"], tmp_path / "o1.snp")
    assert got == golden_bytes("m1_m5.snp") and "Phase 4/4" in log and "Stored to file 54 events" in log
    got, _ = run(["-1", str(tmp_path / "hom_a.ebwt"), "-2", str(tmp_path / "hom_b.ebwt"), "-m", "4", "-v", "1"], tmp_path / "o2.snp")
    assert got == golden_bytes("m2_hom_m4_v1.snp")
    got, _ = run(["-1", str(tmp_path / "hom_m.ebwt"), "-d", str(tmp_path / "hom_m.da"), "-g", "2"], tmp_path / "o3.snp")
    assert got == golden_bytes("m3_hom_g2.snp")
    got, _ = run(["-1", str(tmp_path / "tiny.ebwt")], tmp_path / "o4.snp")     # the reference dies with SIGFPE here; we exit 0
    assert got == b""
    r = subprocess.run([_lib.CLI_PATH, "-1", str(tmp_path / "hom_m.ebwt"), "-d", str(tmp_path / "absent.da"), "-o", str(tmp_path / "o6.snp")], capture_output=True, text=True)
    assert r.returncode == 1 and f"Error: could not read file {tmp_path / 'absent.da'}" in r.stdout
    bad = bytearray(golden_bytes("tiny.ebwt"))
    bad[3] = ord("N")
    (tmp_path / "bad.ebwt").write_bytes(bytes(bad))
    r = subprocess.run([_lib.CLI_PATH, "-1", str(tmp_path / "bad.ebwt"), "-o", str(tmp_path / "o5.snp")], capture_output=True, text=True)
    assert r.returncode == 1 and "read forbidden character 'N' (ASCII code 78)" in r.stdout and '"-t 78"' in r.stdout


# ------------------------------------------------------------------ larger seeded inputs (oracle finishes in seconds)

@pytest.mark.parametrize("seed,extra", [(7, []), (8, ["-m", "4", "-k", "14", "-q", "3"])])
def test_config1_size_mode1(oracle, seed, extra):
    """BASELINE.json configs[0]: 10k x 100 bp reads (n = 1.01 M)."""
    reads, _, _ = sim.diploid_sample(50_000, 10_000, 100, seed=seed)
    e = sim.build_ebwt(reads)
    want, st = oracle.run(e, params=_oracle.Params.from_args(extra))
    got = api.run_one_dataset(e, api.make_params(extra))
    assert want.count(b">") > 100
    assert got == want


def test_medium_modes23(oracle):
    r0, r1 = sim.two_samples(40_000, 8_000, 100, seed=31, het_second=True)
    a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
    m, da = sim.build_merged_ebwt(r0, r1)
    p = _oracle.Params.from_args([])
    want2, _ = oracle.run(a, bwt2=b, params=p)
    want3, _ = oracle.run(m, da=da, params=p)
    assert api.run_two_datasets(a, b) == want2
    assert api.run_two_datasets_da(m, da) == want3
    # the DA rebuilt on the GPU from the two BWTs equals the merged builder's document array
    g1, g2 = api.DnaBwt(a), api.DnaBwt(b)
    fl = api.Flags.build(g1, g2, 16, 30)
    da_bits = np.unpackbits(fl.export()[3].view(np.uint8), bitorder="little")[: len(da)]
    assert np.array_equal(da_bits, (da == ord("1")).astype(np.uint8))


@pytest.mark.parametrize("psi", ["0", "1"], ids=["directory_select", "fl_table"])
@pytest.mark.parametrize("n0,n1", [(8_000, 24_000), (24_000, 8_000)], ids=["second_3x_larger", "second_3x_smaller"])
def test_two_bwts_of_different_sizes_walk_each_in_its_own_index(oracle, monkeypatch, n0, n1, psi):
    """Mode 2 with collections of different sizes (advisor, round 3): a right-context walk in the second BWT must take that
    index's size for the clamp of its select guess -- with the first one's, a larger second index had its correct guesses
    pulled back (dozens of dependent block probes per FL step instead of about one, all counted in requests_right), a
    smaller one was left unguarded.  Same .snp as the oracle, and about two memory requests per FL step (directory + block)."""
    monkeypatch.setenv("EB_PSI", psi)      # "1": each index also brings its own FL table (built with the first job)
    g = sim.random_genome(40_000, 77)
    h2 = sim.mutate_haplotype(g, 77)
    r0, r1 = sim.sample_reads([g], n0, 100, 77), sim.sample_reads([g, h2], n1, 100, 78)
    a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
    p = _oracle.Params.from_args([])
    want, _ = oracle.run(a, bwt2=b, params=p)
    assert want.count(b">") > 20
    pp = api.make_params()
    b1, b2 = api.DnaBwt(a), api.DnaBwt(b)
    fl = api.navigate_two_bwts(b1, b2, pp)
    calls = api.scan_clusters(b1, fl, pp, bwt2=b2)
    assert calls.format_snp(pp, True)[0] == want
    cl = np.frombuffer(calls.arrays()[0].tobytes(), dtype=np.dtype([("begin", "<u8"), ("end", "<u8"), ("first_chain", "<u4"), ("n_slots", "<u2"),
                                                                     ("right_found", "<u2"), ("right_len", "<u4"), ("reserved", "<u4")]))
    steps = int(cl["right_len"].sum())
    assert steps > 1000
    per_step = calls.stats.requests_right / steps
    assert per_step < (2.6 if psi == "0" else 1.1), f"{per_step:.2f} memory requests per FL step: the select's guess is being pulled off its block"


def test_properties_at_larger_size():
    """Size-independent properties on an input too large for per-query oracle checks in the default run:
    rank is monotone, rank(n) sums to n - #TERM, LF intervals tile the F column, every LCP slot is written."""
    reads, _, _ = sim.diploid_sample(400_000, 80_000, 100, seed=99)       # n = 8.08 M
    e = sim.build_ebwt(reads)
    g = api.DnaBwt(e)
    n = len(e)
    counts = [int((e == c).sum()) for c in b"ACGT"]
    assert g.F_arr == [n - sum(counts), n - sum(counts) + counts[0], n - sum(counts) + counts[0] + counts[1], n - counts[3]]
    pos = np.sort(_positions(n, 300_000, 11))
    r = g.parallel_rank(pos)
    assert (np.diff(r.astype(np.int64), axis=0) >= 0).all()
    d = np.diff(pos.astype(np.int64))
    assert (np.diff(r.astype(np.int64), axis=0).sum(axis=1) <= d).all()
    assert [int(x) for x in g.parallel_rank(np.array([n], dtype=np.uint64))[0]] == counts
    full = g.LF(np.array([[0, n]], dtype=np.uint64))[0]
    assert [int(x) for x in full] == [g.F_arr[0], g.F_arr[1], g.F_arr[1], g.F_arr[2], g.F_arr[2], g.F_arr[3], g.F_arr[3], n]
    fl = api.Flags.build(g, None, 16, 30)
    assert fl.stats.lcp_values == n
    thrK, thrR, mn, _ = fl.export()
    assert not (thrR & ~thrK).any()                           # LCP >= 30 implies LCP >= 16
    calls = api.scan_clusters(g, fl, api.make_params())
    assert calls.stats.survivors == len(calls) > 500
    text, cluster_nr, events = calls.format_snp(api.make_params(), False)
    assert events == text.count(b">") and events > 500


def test_long_clusters_take_the_memory_path(oracle):
    """Clusters longer than the 128 positions a lane holds in registers (deep coverage, repeats) go through k_long_runs;
    results must not depend on which path analysed a cluster (modes 1 and 3)."""
    rng = _rng(77)
    unit = sim.BASES[rng.integers(0, 4, size=400)]
    g = np.concatenate([sim.BASES[rng.integers(0, 4, size=1500)], unit, unit, unit, sim.BASES[rng.integers(0, 4, size=1500)]])
    h2 = g.copy()
    for pos in (700, 1700, 2100, 2500, 3300):
        h2[pos] = ord("A") if g[pos] != ord("A") else ord("C")
    reads = sim.sample_reads([g, h2], 9000, 100, seed=78)           # ~200x: every locus cluster is > 128 rows
    e = sim.build_ebwt(reads)
    p = _oracle.Params.from_args([])
    want, st = oracle.run(e, params=p)
    assert api.run_one_dataset(e) == want
    o1 = oracle.index(e)
    of = _oracle.OracleFlags(oracle, o1, None, p.K, p.k_right)
    rec = of.scan(p, 0, len(e), capture=True)
    cl = np.frombuffer(rec["clusters"].tobytes(), dtype=np.uint64).reshape(-1, 4)
    assert ((cl[:, 1] - cl[:, 0]) > 130).sum() > 10              # long surviving clusters exist
    g1 = api.DnaBwt(e)
    calls = api.scan_clusters(g1, api.Flags.from_host(len(e), *of.export()), api.make_params())
    assert bytes(calls.blob()) == _oracle.pack_calls(rec, p.k_left, p.k_right)
    # same reads as two samples through a document array (mode 3)
    r0, r1 = reads[:4500], reads[4500:]
    m, da = sim.build_merged_ebwt(r0, r1)
    want3, _ = oracle.run(m, da=da, params=p)
    assert api.run_two_datasets_da(m, da) == want3


DEGENERATE = {
    "empty": [],
    "one_base": [b"A"],
    "same_base_x4": [b"A"] * 4,
    "one_read": [b"ACGTTGCAACGT"],
    "identical_reads": [b"ACGTACGGTCAATCGGCTAAGCTTAGCCTAGGATCCA"] * 12,
    "length_one_reads": [b"A", b"C", b"G", b"T"] * 5,
    "homopolymer": [b"A" * 60] * 8 + [b"A" * 59 + b"C"] * 8,
}


@pytest.mark.parametrize("name", list(DEGENERATE), ids=list(DEGENERATE))
def test_degenerate_inputs(oracle, name):
    """Empty, single-symbol, duplicate-only and homopolymer collections: index, flag vectors and .snp against the oracle."""
    reads = DEGENERATE[name]
    e = sim.build_ebwt(reads).tobytes() if reads else b""
    p = _oracle.Params.from_args([])
    want, st = oracle.run(e, params=p)
    got = api.run_one_dataset(e)
    assert got == want
    g = api.DnaBwt(e)
    assert g.size() == len(e)
    o = oracle.index(e)
    assert g.F_arr == [int(x) for x in o.F()]
    pos = np.arange(len(e) + 1, dtype=np.uint64)
    assert np.array_equal(g.parallel_rank(pos), o.rank4(pos))
    gf = api.Flags.build(g, None, p.K, p.k_right)
    of = _oracle.OracleFlags(oracle, o, None, p.K, p.k_right)
    for a, b in zip(gf.export()[:3], of.export()[:3]):
        assert np.array_equal(a, b)
    oc = of.counters()
    assert (gf.stats.leaves, gf.stats.nodes, gf.stats.lcp_values) == (oc["leaves"], oc["nodes"], oc["lcp_values"])


def test_device_image_equals_host_blob(dip):
    """A device-only job (EB_SCAN_DEVICE_ONLY) leaves the call set in HBM as one contiguous image: byte-identical to the
    blob a normal job downloads, for a full range, a shard and an empty range."""
    e, g, _ = dip
    n = len(e)
    p = api.make_params()
    fl = api.navigate_one_bwt(g, p)
    for begin, end in [(0, n), (n // 3 // 64 * 64, n // 3 * 2 // 64 * 64), (64, 64)]:
        host = api.scan_clusters_begin(g, fl, p, begin=begin, end=end).finish()
        want = bytes(host.blob())
        job = api.scan_clusters_begin(g, fl, p, begin=begin, end=end, device_only=True, wire=(begin == 0))   # packed by the job / on demand
        job.wait()
        ptr, length = job.device_blob()
        assert length == len(want)
        got = np.empty(length, dtype=np.uint8)
        sim._hip_copy_to_host(got, ptr, length)
        wptr, wlen = job.device_wire()                           # compact wire image of the same call set
        wire = np.empty(wlen, dtype=np.uint8)
        sim._hip_copy_to_host(wire, wptr, wlen)
        done = job.finish()
        assert done._h is None and done.stats.survivors == len(host)
        assert got.tobytes() == want
        assert wlen <= length and (len(host) == 0 or wlen < 0.75 * length)
        from_wire = api.Calls.unpack(wire.tobytes())
        assert bytes(from_wire.blob()) == want                   # unpacks to the very same records
        back = api.Calls.unpack(got.tobytes())
        assert back.format_snp(p, False)[0] == host.format_snp(p, False)[0]


def test_pipelined_jobs_end_in_any_order_and_on_another_thread(dip):
    """EB_SCAN_PIPELINED: jobs are enqueued whole by _begin and may be waited for / ended / asked for their images in any order,
    from another host thread than the one that keeps beginning jobs.  Every order gives the blob of a plain job."""
    import threading
    e, g, _ = dip
    n = len(e)
    p = api.make_params()
    fl = api.navigate_one_bwt(g, p)
    cuts = [(0, n), (n // 3 // 64 * 64, n // 3 * 2 // 64 * 64), (64, 64), (n // 2 // 64 * 64, n)]

    def blob_of(job):                      # the view returned by blob() is only valid while its Calls object lives
        calls = job.finish()
        return bytes(calls.blob())
    want = [blob_of(api.scan_clusters_begin(g, fl, p, begin=b, end=e_)) for b, e_ in cuts]
    # ended at once; ended after the next begin; ended in reverse order
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [1, 0, 3, 2]):
        jobs = [api.scan_clusters_begin(g, fl, p, begin=b, end=e_, pipelined=True) for b, e_ in cuts]
        got = {}
        for k in order:
            got[k] = blob_of(jobs[k])
        assert [got[k] for k in range(len(cuts))] == want
    for k, (b, e_) in enumerate(cuts):
        assert blob_of(api.scan_clusters_begin(g, fl, p, begin=b, end=e_, pipelined=True)) == want[k]
    # device images + wire, handed over on another thread while this one keeps beginning: a job is passed on either at once
    # (the helper may send its last part out itself, concurrently with the next _begin) or after the next _begin returned
    import queue
    for lag in (0, 1):
        q, out = queue.Queue(), []

        def hand_over():
            while True:
                job = q.get()
                if job is None:
                    return
                job.wait()
                ptr, length = job.device_wire()
                wire = np.empty(length, dtype=np.uint8)
                sim._hip_copy_to_host(wire, ptr, length)
                job.finish()
                unpacked = api.Calls.unpack(wire.tobytes())
                out.append(bytes(unpacked.blob()))
        th = threading.Thread(target=hand_over)
        th.start()
        held = []
        for b, e_ in cuts * 4:
            held.append(api.scan_clusters_begin(g, fl, p, begin=b, end=e_, device_only=True, wire=True, pipelined=True))
            while len(held) > lag:
                q.put(held.pop(0))
        for job in held:
            q.put(job)
        q.put(None)
        th.join()
        assert out == want * 4
    # a cancelled (never waited for) pipelined job, and one whose index sees no further begin
    j = api.scan_clusters_begin(g, fl, p, pipelined=True)
    del j
    assert blob_of(api.scan_clusters_begin(g, fl, p)) == want[0]


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m2_het_default", "m3_het_default")], ids=lambda c: c["name"])
@pytest.mark.parametrize("tiles", ["1", "4", "32"])
def test_batch_size_of_the_ordered_hand_over_does_not_change_the_call_set(oracle, monkeypatch, case, tiles):
    """The scan hands its survivors over in BWT-ordered batches of 128 tiles (2 M positions) that shrink by themselves when one
    overflows; EB_SCAN_ORDERED=<tiles> starts with smaller ones.  Records, counters and .snp are those of the default -- whole
    range and an odd slice, ungated and gated context walks -- and read data needs no second run at any size."""
    p, e1, e2, da = _flag_case(oracle, case)
    pp = api.make_params(case["args"])
    g1 = api.DnaBwt(e1, term=p.term)
    g2 = api.DnaBwt(e2, term=p.term) if e2 is not None else None
    gf = api.Flags.build(g1, g2, p.K, p.k_right)
    if da is not None:
        gf.set_da(da)
    n = g1.n + (g2.n if g2 is not None else 0)
    # (the batch size is remembered per index: the small batches get indexes of their own)
    h1 = api.DnaBwt(e1, term=p.term)
    h2 = api.DnaBwt(e2, term=p.term) if e2 is not None else None
    for gated in (False, True):                 # the walk: for every cluster (small call sets), gated (large call sets)
        for b, e in ((0, n), (n // 5 + 3, n - n // 7)):
            monkeypatch.setenv("EB_CONTEXT", "gate=0" if gated else "gate=524288")
            monkeypatch.delenv("EB_SCAN_ORDERED", raising=False)
            plain = api.scan_clusters(g1, gf, pp, bwt2=g2, begin=b, end=e)
            monkeypatch.setenv("EB_SCAN_ORDERED", tiles)
            cut = api.scan_clusters(h1, gf, pp, bwt2=h2, begin=b, end=e)
            assert plain.stats.batch_tiles == 128 and cut.stats.batch_tiles == int(tiles)
            assert plain.stats.scan_reruns == 0 and cut.stats.scan_reruns == 0
            assert bytes(cut.blob()) == bytes(plain.blob())
            a, c = plain.stats, cut.stats
            assert (a.runs_closed, a.clusters, a.bases_in_runs, a.survivors, a.chains, list(a.hist)) == \
                   (c.runs_closed, c.clusters, c.bases_in_runs, c.survivors, c.chains, list(c.hist))
            if (b, e) == (0, n) and not gated:
                assert cut.format_snp(pp, case["mode"] != 1)[0] == golden_bytes(case["snp"])


def _bits(rng, n, p_one):
    return rng.random(n) < p_one


def _pack_bits(bits):
    pad = (-len(bits)) % 64
    return np.packbits(np.concatenate([bits, np.zeros(pad, dtype=bool)]), bitorder="little").view(np.uint64).copy()


_STRESS_RERUNS, _STRESS_TILES = {}, {}
_STRESS = {
    # name: (n, P(symbol is TERM), flag pattern)
    "dense_runs_of_20": (200_000, 0.02, lambda r, n: _bits(r, n, 0.95)),
    "short_runs": (200_000, 0.02, lambda r, n: _bits(r, n, 0.6)),
    "nine_clusters_per_word": (120_000, 0.02, lambda r, n: (np.arange(n) % 7) != 6),
    "long_runs_150": (300_000, 0.02, lambda r, n: (np.arange(n) % 151) != 0),
    "runs_across_many_words": (300_000, 0.0, lambda r, n: _bits(r, n, 0.9993)),
    "one_giant_run": (150_000, 0.01, lambda r, n: (np.arange(n) != 70_001) & (np.arange(n) > 3)),
    "word_aligned_runs": (130_000, 0.05, lambda r, n: (np.arange(n) % 64) != 63),
    "two_word_runs": (130_000, 0.05, lambda r, n: (np.arange(n) % 128) != 0),
}


@pytest.mark.parametrize("scan_form", ["default", "overlap3", "one_tile_batches", "unordered"])
@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name", sorted(_STRESS))
def test_scan_on_adversarial_flag_patterns(oracle, monkeypatch, name, mode, scan_form):
    """Phase 4 on inputs the navigation would never produce: a random symbol string (every cluster mixes letters, so all of
    them reach the queues, the analysis and the survivor staging overflow) under flag patterns that pin the corner
    cases of the run logic -- runs ending on word boundaries, spanning exactly two words, spanning hundreds, nine
    clusters in one word.  The scan only reads these vectors (ebwt2InDel.cpp:1609-1655); records, counters and
    CLUST_SIZES must equal the oracle's, whole range and cut at odd positions."""
    if scan_form == "overlap3":                # the scan that shares the CUs with the context kernels: 3 workgroups per CU, tile loads one tile ahead
        monkeypatch.setenv("EB_SCAN_OVERLAP", "3")
    elif scan_form == "one_tile_batches":      # the ordered hand-over from its smallest batches on
        monkeypatch.setenv("EB_SCAN_ORDERED", "1")
    elif scan_form == "unordered":             # the fallback at once
        monkeypatch.setenv("EB_SCAN_ORDERED", "0")
    n, p_term, pattern = _STRESS[name]
    rng = _rng(sum(name.encode()) * 7 + mode)
    sym = rng.choice(np.frombuffer(b"ACGT#", dtype=np.uint8), size=n, p=[(1 - p_term) / 4] * 4 + [p_term])
    if p_term == 0.0:
        sym[:8] = ord("#")                                       # F_A > 0: row 0 is a TERM row like in any eBWT
    thrK = _pack_bits(pattern(rng, n))
    thrR = thrK & _pack_bits(_bits(rng, n, 0.5))
    minima = _pack_bits(_bits(rng, n, 0.01))
    da_bits = _bits(rng, n, 0.5) if mode != 1 else None
    da = _pack_bits(da_bits) if mode != 1 else None
    args = ["-m", "2"] if name == "short_runs" else []
    p, pp = _oracle.Params.from_args(args), api.make_params(args)
    o2 = g2 = None
    if mode == 2:
        # two symbol strings whose merge by the DA is `sym`: the scan must see the same rows as in mode 3
        sym1, sym2 = sym[~da_bits].copy(), sym[da_bits].copy()
        sym1[:4] = ord("#")
        sym2[:4] = ord("#")
        o1, o2 = oracle.index(sym1, p.term), oracle.index(sym2, p.term)
        g1, g2 = api.DnaBwt(sym1, term=p.term), api.DnaBwt(sym2, term=p.term)
    else:
        o1 = oracle.index(sym, p.term)
        g1 = api.DnaBwt(sym, term=p.term)
    of = _oracle.OracleFlags(oracle, o1, o2, vectors=(thrK, thrR, minima, da))
    gf = api.Flags.from_host(n, thrK, thrR, minima, da)
    seen, reruns, tiles = [], [], []
    for cuts in ([0, n], [0, 64 * 700 + 1, n // 2 + 17, n - 5, n]):
        for b, e in zip(cuts[:-1], cuts[1:]):
            want = of.scan(p, b, e, capture=True)
            seen.append(want["count"])
            calls = api.scan_clusters(g1, gf, pp, bwt2=g2, begin=b, end=e)
            cl, chains, left, right = calls.arrays()
            st = calls.stats
            reruns.append(st.scan_reruns)
            tiles.append(st.batch_tiles)
            assert dict(runs_closed=st.runs_closed, clusters=st.clusters, bases_in_runs=st.bases_in_runs,
                        survivors=st.survivors, chains=st.chains) == want["stats"]
            assert np.array_equal(np.array(list(st.hist), dtype=np.uint64), want["hist"])
            assert np.array_equal(cl, want["clusters"])
            assert np.array_equal(chains, want["chains"])
            assert np.array_equal(left, want["left"])
            assert np.array_equal(right, want["right"])
    assert seen[0] == sum(seen[1:]) and seen[0] > 0              # the cuts partition the clusters; the input is not trivial
    if scan_form == "default":
        _STRESS_RERUNS[(name, mode)] = max(reruns)
        _STRESS_TILES[(name, mode)] = min(tiles)
    elif scan_form == "unordered":
        assert max(tiles) == 0


def test_adversarial_patterns_reach_the_smaller_batches_and_the_unordered_scan():
    """The scan hands its survivors over in BWT-ordered batches of 512 slots; a batch that overflows makes the job run again with
    smaller batches, and the unordered way when not even one-tile batches hold it (eb_scan_stats.scan_reruns, .batch_tiles).  Read
    data never does that, the dense patterns above must -- or both fallbacks would go untested."""
    if not _STRESS_RERUNS:
        pytest.skip("runs after test_scan_on_adversarial_flag_patterns")
    assert max(_STRESS_RERUNS.values()) > 0, _STRESS_RERUNS
    assert 0 in _STRESS_TILES.values(), _STRESS_TILES                       # some pattern overflows one-tile batches too


def test_batches_shrink_to_the_density_they_meet(monkeypatch):
    """A range long enough for a workgroup to own several tiles (2048 workgroups x 4 tiles of 16 384 positions), with a surviving
    cluster every 64 positions: four-tile batches hold 1024 of them in 512 slots, so the first job spills, is done again with
    one-tile batches (256 each) and the index remembers that size -- the call set is the unordered hand-over's, byte for byte."""
    n = 2048 * 4 * 16384 + 777
    rng = _rng(77)
    sym = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    sym[:8] = ord("#")
    thrK = _pack_bits((np.arange(n) % 64) < 20)
    thrR = thrK.copy()
    minima = np.zeros_like(thrK)
    p = api.make_params()
    g = api.DnaBwt(sym)
    gf = api.Flags.from_host(n, thrK, thrR, minima, None)
    first = api.scan_clusters(g, gf, p)
    assert first.stats.scan_reruns >= 1 and 0 < first.stats.batch_tiles < 4, (first.stats.scan_reruns, first.stats.batch_tiles)
    assert first.stats.survivors > n // 64 * 0.9
    again = api.scan_clusters(g, gf, p)                      # the index remembers
    assert again.stats.scan_reruns == 0 and again.stats.batch_tiles == first.stats.batch_tiles
    jobs = [api.scan_clusters_begin(g, gf, p, pipelined=True) for _ in range(3)]
    monkeypatch.setenv("EB_SCAN_ORDERED", "0")
    g2 = api.DnaBwt(sym)
    plain = api.scan_clusters(g2, gf, p)
    assert plain.stats.batch_tiles == 0
    want = bytes(plain.blob())
    assert bytes(first.blob()) == want and bytes(again.blob()) == want
    assert [bytes(j.finish().blob()) for j in jobs] == [want] * 3
    a, c = first.stats, plain.stats
    assert (a.runs_closed, a.clusters, a.bases_in_runs, a.survivors, a.chains, list(a.hist)) == \
           (c.runs_closed, c.clusters, c.bases_in_runs, c.survivors, c.chains, list(c.hist))
    del first, again, plain, jobs
    # three jobs of a pipelining caller in flight on a fresh index when the first of them turns out to have spilled: each is done again
    monkeypatch.delenv("EB_SCAN_ORDERED")
    g3 = api.DnaBwt(sym)
    jobs = [api.scan_clusters_begin(g3, gf, p, pipelined=True) for _ in range(3)]
    outs = [j.finish() for j in jobs]
    assert [bytes(o.blob()) for o in outs] == [want] * 3
    assert outs[0].stats.scan_reruns >= 1 and api.scan_clusters(g3, gf, p).stats.scan_reruns == 0


def test_ordered_and_unordered_scan_give_the_same_call_set(dip, monkeypatch):
    """EB_SCAN_ORDERED=0 (one survivor list put in order by position buckets) against the default (ordered batches) on
    read data: same bytes, and neither needs a second run."""
    e, g, _ = dip
    n = len(e)
    p = api.make_params()
    fl = api.navigate_one_bwt(g, p)
    for b, e_ in ((0, n), (n // 3, n - n // 5), (64, 64)):
        monkeypatch.delenv("EB_SCAN_ORDERED", raising=False)
        a = api.scan_clusters(g, fl, p, begin=b, end=e_)
        monkeypatch.setenv("EB_SCAN_ORDERED", "0")
        c = api.scan_clusters(g, fl, p, begin=b, end=e_)
        assert bytes(a.blob()) == bytes(c.blob())
        assert a.stats.scan_reruns == 0 and c.stats.scan_reruns == 0


@pytest.mark.skipif(not os.path.exists(_oracle.REF_BIN), reason="reference binary did not travel with the tree")
@pytest.mark.parametrize("seed,extra", [(301, []), (302, ["-m", "4", "-k", "18"]), (303, ["-L", "36", "-R", "28", "-g", "4", "-q", "2"]),
                                        (304, ["-L", "70", "-R", "66", "-k", "12"]),          # contexts of several words
                                        (305, ["-m", "1", "-v", "1", "-k", "9"]),             # every run of two rows is a cluster
                                        (306, ["-c", "6", "-q", "2", "-m", "2", "-g", "1"])])
def test_fresh_inputs_against_the_reference_binary(tmp_path, seed, extra):
    """Not via fixtures and not via the oracle: the unmodified reference (oracle/_ref, built in the build container from
    the upstream sources; git-ignored, travels with the tree) runs here on fresh seeded inputs in all three modes, and
    the GPU path must write the same bytes."""
    def ref(args):
        out = tmp_path / "ref.snp"
        rc = subprocess.run([_oracle.REF_BIN] + args + extra + ["-o", str(out)], stdout=subprocess.DEVNULL).returncode
        assert rc == 0
        return out.read_bytes()

    p = api.make_params(extra)
    reads, _, _ = sim.diploid_sample(30_000, 7_000, 100, seed=seed, snp_every=350, indel_every=900)
    e = sim.build_ebwt(reads)
    (tmp_path / "x.ebwt").write_bytes(e.tobytes())
    want = ref(["-1", str(tmp_path / "x.ebwt")])
    assert want.count(b">") > 20 and api.run_one_dataset(e, p) == want
    r0, r1 = sim.two_samples(20_000, 4_500, 90, seed=seed + 50, het_second=True, snp_every=250, indel_every=700)
    a, b = sim.build_ebwt(r0), sim.build_ebwt(r1)
    m, da = sim.build_merged_ebwt(r0, r1)
    for name, x in (("a", a), ("b", b), ("m", m), ("da", da)):
        (tmp_path / name).write_bytes(x.tobytes())
    want2 = ref(["-1", str(tmp_path / "a"), "-2", str(tmp_path / "b")])
    want3 = ref(["-1", str(tmp_path / "m"), "-d", str(tmp_path / "da")])
    assert want2.count(b">") > 20 and api.run_two_datasets(a, b, p) == want2
    assert want3.count(b">") > 20 and api.run_two_datasets_da(m, da, p) == want3


@pytest.mark.skipif(not os.path.exists(_oracle.REF_BIN), reason="reference binary did not travel with the tree")
@pytest.mark.parametrize("rc", [False, True], ids=["errors", "errors_rc"])
def test_reads_with_errors_against_the_reference_binary_at_60m_positions(tmp_path, rc):
    """The reference's documented kind of input (README.md:88-100) straight through the unmodified reference binary on the test box:
    400 k x 150 bp reads with 1 % substitution errors (and, second case, 200 k such reads plus their reverse complements), 60 M
    positions built by the GPU generator -- the GPU path must write the reference's bytes (round 4's fresh inputs were error-free;
    the goldens with errors stop at 0.4 M positions)."""
    n_reads, L = (200_000, 150) if rc else (400_000, 150)
    g = sim.random_genome(int(n_reads * (2 if rc else 1) * L / 30), 911)
    h2 = sim.mutate_haplotype(g, 911)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 911)
    d_reads, d_starts = sim.error_sample_gpu(text, starts, L, 911, 10_000, rc=rc)
    dev = sim.build_ebwt_gpu_slabbed(int(d_reads.data_ptr()), int(d_starts.data_ptr()), L, n_reads=int(d_starts.numel()), text_len=int(d_reads.numel()))
    e, _ = dev.to_host()
    dev.free()
    del d_reads, d_starts
    assert len(e) == 400_000 * 151
    pth, out = tmp_path / "x.ebwt", tmp_path / "ref.snp"
    e.tofile(pth)
    r = subprocess.run([_oracle.REF_BIN, "-1", str(pth), "-o", str(out)], stdout=subprocess.DEVNULL, timeout=600)
    assert r.returncode == 0
    want = out.read_bytes()
    assert want.count(b">") > 2000
    assert api.run_one_dataset(e, api.make_params()) == want
    # and the executable on the same file
    got = tmp_path / "gpu.snp"
    r = subprocess.run([_lib.CLI_PATH, "-1", str(pth), "-o", str(got)], capture_output=True, text=True)
    assert r.returncode == 0 and got.read_bytes() == want


def test_wire_image_falls_back_to_full_width_records(oracle):
    """A cluster of more than 65535 rows does not fit the tight wire image (16-bit length): the job then packs the
    full-width one, and both unpack to the very records of the host blob."""
    n = 150_000
    rng = _rng(77)
    sym = rng.choice(np.frombuffer(b"ACGT#", dtype=np.uint8), size=n, p=[0.2475] * 4 + [0.01])
    bits = (np.arange(n) != 70_001) & (np.arange(n) > 3)          # two runs of ~70 k and ~80 k rows
    thrK = _pack_bits(bits)
    thrR = thrK & _pack_bits(_bits(rng, n, 0.5))
    minima = _pack_bits(np.zeros(n, dtype=bool))
    p = api.make_params()
    g = api.DnaBwt(sym)
    fl = api.Flags.from_host(n, thrK, thrR, minima)
    host = api.scan_clusters_begin(g, fl, p).finish()
    assert len(host) >= 1 and max(int(c.end - c.begin) for c in _clusters(host)) > 65535
    for at_begin in (True, False):
        job = api.scan_clusters_begin(g, fl, p, device_only=True, wire=at_begin)
        job.wait()
        ptr, length = job.device_wire()
        wire = np.empty(length, dtype=np.uint8)
        sim._hip_copy_to_host(wire, ptr, length)
        job.finish()
        assert wire[:8].tobytes() == b"EBCALLS3"
        assert bytes(api.Calls.unpack(wire.tobytes()).blob()) == bytes(host.blob())
    # and an ordinary call set takes the tight image
    e = golden_bytes("dip.ebwt")
    g2 = api.DnaBwt(e)
    fl2 = api.navigate_one_bwt(g2, p)
    job = api.scan_clusters_begin(g2, fl2, p, device_only=True, wire=True)
    job.wait()
    ptr, length = job.device_wire()
    wire = np.empty(length, dtype=np.uint8)
    sim._hip_copy_to_host(wire, ptr, length)
    job.finish()
    assert wire[:8].tobytes() == b"EBCALLS4"
    assert bytes(api.Calls.unpack(wire.tobytes()).blob()) == bytes(api.scan_clusters(g2, fl2, p).blob())


def _clusters(calls):
    v = calls.view()
    return [v.clusters[i] for i in range(v.count)]


@pytest.mark.skipif(not os.path.exists(_oracle.REF_BIN), reason="reference binary did not travel with the tree")
@pytest.mark.parametrize("extra", [["-m", "12"], ["-m", "25"], ["-m", "40", "-k", "20"], ["-m", "33", "-q", "3"]], ids=lambda x: "".join(x))
def test_large_min_coverage_against_the_reference_binary(tmp_path, extra):
    """-m sets the shortest cluster analysed (2 x mcov rows): with 80 of them a cluster never fits one 64-row word of the
    scan.  A 200x sample, against the reference binary."""
    reads, _, _ = sim.diploid_sample(20_000, 40_000, 100, seed=11, snp_every=300, indel_every=800)
    e = sim.build_ebwt(reads)
    (tmp_path / "x.ebwt").write_bytes(e.tobytes())
    out = tmp_path / "ref.snp"
    rc = subprocess.run([_oracle.REF_BIN, "-1", str(tmp_path / "x.ebwt"), "-o", str(out)] + extra, stdout=subprocess.DEVNULL).returncode
    want = out.read_bytes()
    assert rc == 0 and want.count(b">") >= 2
    assert api.run_one_dataset(e, api.make_params(extra)) == want
