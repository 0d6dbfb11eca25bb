"""Host logic of the product (C++ event logic + KisSNP2 writer + call-set container, csrc/eb_writer.cpp)
checked on the CPU: cluster records captured from the oracle's phase 4 are fed through the product's
eb_calls_unpack / eb_format_snp and must reproduce the reference-written golden .snp byte for byte --
whole range and sharded (records gathered shard by shard with a running cluster_nr)."""
import pytest

from ebwt2indel_amd import api
from tests import _oracle
from tests.conftest import CASES, CASE_IDS, golden_bytes


def _setup(oracle, case):
    p = _oracle.Params.from_args(case["args"])
    inp = case["inputs"]
    x1 = oracle.index(golden_bytes(inp["bwt1"]), p.term)
    x2 = oracle.index(golden_bytes(inp["bwt2"]), p.term) if case["mode"] == 2 else None
    da = golden_bytes(inp["da"]) if case["mode"] == 3 else None
    fl = _oracle.OracleFlags(oracle, x1, x2, p.K, p.k_right, da)
    n = x1.n + (x2.n if x2 is not None else 0)
    return p, fl, n


@pytest.mark.parametrize("case", CASES, ids=CASE_IDS)
def test_writer_reproduces_golden(oracle, case):
    p, fl, n = _setup(oracle, case)
    rec = fl.scan(p, 0, n, capture=True)
    assert rec["text"] == golden_bytes(case["snp"])          # the oracle's own phase-4 entry point
    pp = api.make_params(case["args"])
    calls = api.Calls.unpack(_oracle.pack_calls(rec, p.k_left, p.k_right))
    assert len(calls) == rec["count"]
    text, cluster_nr, events = calls.format_snp(pp, case["mode"] != 1)
    assert text == golden_bytes(case["snp"])
    assert cluster_nr == rec["cluster_nr"]
    if case["mode"] == 1:
        assert events == rec["events"] == case["records"]
    assert api.Calls.unpack(calls.pack()).format_snp(pp, case["mode"] != 1)[0] == text   # pack/unpack round trip


@pytest.mark.parametrize("threads", ["2", "7"])
@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m1_edge_default", "m2_het_default", "m3_het_default")],
                         ids=lambda c: c["name"])
def test_writer_threads_do_not_change_the_text(oracle, monkeypatch, case, threads):
    """Large call sets are formatted in pieces by host threads and numbered when the pieces are joined: same bytes,
    same counters, also when the caller continues a running cluster_nr."""
    p, fl, n = _setup(oracle, case)
    rec = fl.scan(p, 0, n, capture=True)
    pp = api.make_params(case["args"])
    calls = api.Calls.unpack(_oracle.pack_calls(rec, p.k_left, p.k_right))
    monkeypatch.setenv("EB_WRITER_THREADS", "1")
    want = calls.format_snp(pp, case["mode"] != 1, 41, 7)
    monkeypatch.setenv("EB_WRITER_THREADS", threads)
    assert calls.format_snp(pp, case["mode"] != 1, 41, 7) == want
    monkeypatch.delenv("EB_WRITER_THREADS")
    assert calls.format_snp(pp, case["mode"] != 1)[0] == golden_bytes(case["snp"])


@pytest.mark.parametrize("threads", ["1", "3", "16"])
@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m2_het_default", "m3_het_default")], ids=lambda c: c["name"])
def test_write_snp_puts_every_piece_at_its_offset_of_the_file(oracle, monkeypatch, tmp_path, case, threads):
    """eb_write_snp: every host thread finishes its piece and writes it at its own offset (no joined text): the file is the
    reference's; a second call that appends continues the file and the cluster numbers (the multi-GPU executable writes rank by rank)."""
    p, fl, n = _setup(oracle, case)
    pp = api.make_params(case["args"])
    two = case["mode"] != 1
    monkeypatch.setenv("EB_WRITER_THREADS", threads)
    out = tmp_path / "o.snp"
    out.write_bytes(b"stale bytes that a fresh write must not keep " * 4000)
    rec = fl.scan(p, 0, n, capture=True)
    calls = api.Calls.unpack(_oracle.pack_calls(rec, p.k_left, p.k_right))
    cn, ev = calls.write_snp(pp, two, str(out))
    assert out.read_bytes() == golden_bytes(case["snp"]) and cn == rec["cluster_nr"]
    # two shards, the second appended
    cut = n // 2
    cn, ev = 1, 0
    for i, (b, e) in enumerate(((0, cut), (cut, n))):
        r = fl.scan(p, b, e, capture=True)
        cn, ev = api.Calls.unpack(_oracle.pack_calls(r, p.k_left, p.k_right)).write_snp(pp, two, str(out), append=i > 0, cluster_nr=cn, events=ev)
    assert out.read_bytes() == golden_bytes(case["snp"]) and cn == rec["cluster_nr"]
    # appending to a file that does not exist yet creates it
    fresh = tmp_path / "fresh.snp"
    calls.write_snp(pp, two, str(fresh), append=True)
    assert fresh.read_bytes() == golden_bytes(case["snp"])
    with pytest.raises(api.EbError, match="cannot open"):
        calls.write_snp(pp, two, str(tmp_path / "no_such_dir" / "o.snp"))


@pytest.mark.parametrize("case", [c for c in CASES if c["name"] in ("m1_default", "m1_edge_default", "m2_het_default", "m3_hom_default")],
                         ids=lambda c: c["name"])
@pytest.mark.parametrize("shards", [2, 5])
def test_sharded_scan_concatenates(oracle, case, shards):
    p, fl, n = _setup(oracle, case)
    pp = api.make_params(case["args"])
    cuts = [n * i // shards for i in range(shards + 1)]
    text, cluster_nr, events, stats = b"", 1, 0, None
    for i in range(shards):
        rec = fl.scan(p, cuts[i], cuts[i + 1], capture=True)
        calls = api.Calls.unpack(_oracle.pack_calls(rec, p.k_left, p.k_right))
        t, cluster_nr, events = calls.format_snp(pp, case["mode"] != 1, cluster_nr, events)
        text += t
        stats = rec["stats"] if stats is None else {k: stats[k] + v for k, v in rec["stats"].items()}
    assert text == golden_bytes(case["snp"])
    whole = fl.scan(p, 0, n, capture=True)
    assert stats == whole["stats"]                            # shards partition the clusters exactly
