"""The BASELINE configurations at full size through the CPU oracle, on the GPU box's host threads: what the `-m gpu` tests of
tests/test_gpu_fullsize.py hold the product to (profiles/r03_fullsize_<config>.json).

The unmodified reference needs ~45 min for 7.55 G positions (single thread, 2.9 M positions/s at DRAM-resident sizes:
SURVEY.md section 6), more than a GPU-box call may last; its word past 2^32 is taken on the replicated input of
scripts/ref_past_2p32.py.  Here the ORACLE -- pinned to the reference by the golden fixtures, by the fresh-input runs
against the reference binary and, past 2^32, by tests/test_gpu_superblocks.py -- walks the very inputs the tests regenerate:

  part "traversal": threaded index + traversal (orc_navigate_one_mt / orc_navigate_two_mt: same expansion rule, subtrees
                    over threads; bit-identical to the serial, reference-ordered code on the fixtures,
                    tests/test_oracle_threads.py) -> counters and the SHA-256 of each bit vector, compared with the GPU's;
  part "phase4":    phase 4 over the WHOLE range in all three modes (round 2 compared modes 2 / 3 on windows only): the
                    range is cut into shards that the host threads scan side by side, every shard numbering its clusters
                    from 1, and the texts are renumbered into the single pass's running cluster_nr
                    (tests/_fullsize.py::renumber; tests/test_oracle_threads.py holds that to a one-piece scan) -> SHA-256
                    of the whole .snp, compared with the GPU's.  It runs on the GPU's bit vectors, whose SHA-256 it
                    records: they must equal the ones part "traversal" recorded for the oracle's own (merge checks it).

    gpurun --timeout 1200 -- 'python scripts/fullsize_check.py config2 all     > gpurun_out/fs2.log 2>&1'
    gpurun --timeout 1200 -- 'python scripts/fullsize_check.py config4 traversal > gpurun_out/fs4t.log 2>&1'
    gpurun --timeout 1200 -- 'python scripts/fullsize_check.py config4 phase4a   > gpurun_out/fs4a.log 2>&1'   # [0, cut)
    gpurun --timeout 1200 -- 'python scripts/fullsize_check.py config4 phase4b   > gpurun_out/fs4b.log 2>&1'   # [cut, n); "phase4": the whole range in one call (config3)
    python scripts/fullsize_check.py config4 merge      # gpurun_out/r03_fullsize_config4_*.json -> profiles/r03_fullsize_config4.json
    (also: python scripts/fullsize_check.py 50000000 x10 -- the replicated input towards configs[4], as in round 2)
"""
import hashlib
import json
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ebwt2indel_amd import api, sim  # noqa: E402
from ebwt2indel_amd import distributed as ebd  # noqa: E402
from tests import _fullsize as fs  # noqa: E402
from tests import _oracle  # noqa: E402

NAME = sys.argv[1] if len(sys.argv) > 1 else "config2"
PART = sys.argv[2] if len(sys.argv) > 2 else "all"
READS = int(NAME) if NAME.isdigit() else 0
COPIES = int(PART[1:]) if PART.startswith("x") and PART[1:].isdigit() else 0   # "x10": every read 10 times (sim.replicated_ebwt_gpu)
READ_LEN, COVERAGE, SEED = 150, 30.0, 1
try:
    _cpus = len(os.sched_getaffinity(0))
except AttributeError:
    _cpus = os.cpu_count() or 8
THREADS = int(os.environ.get("EB_CHECK_THREADS", str(min(_cpus, 64))))


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


class Heartbeat:
    """A line a minute while a long C call runs (ctypes releases the GIL): the GPU box kills silent commands."""

    def __init__(self, what):
        self.what, self.stop = what, threading.Event()

    def __enter__(self):
        self.t0 = time.time()

        def beat():
            while not self.stop.wait(60):
                log(f"... {self.what}: {time.time() - self.t0:.0f}s")
        self.th = threading.Thread(target=beat, daemon=True)
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.seconds = time.time() - self.t0
        log(f"{self.what}: {self.seconds:.1f}s")


def out_path(name, part):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    return os.path.join(ROOT, "gpurun_out", os.path.basename(fs.profile_path(name))[:-5] + f"_{part}.json")


def oracle_indexes(orc, inp):
    with Heartbeat(f"oracle index, {THREADS} threads"):
        idx = [orc.index(h, threads=THREADS) for h in inp.hosts]
    assert all(i.ok for i in idx)
    return idx


def select_boundary_check(orc, idx, per_side=2000):
    """The oracle's directory select against the reference's bisection where 64-bit counts matter: the occurrences on both sides
    of every superblock boundary of every index, plus random ones (advisor, round 3).  Returns what goes into the record."""
    import ctypes
    checked, bad, rnd = 0, 0, 0
    with Heartbeat("oracle select: directory form against the reference's bisection around every superblock boundary"):
        for i in idx:
            c = ctypes.c_uint64(0)
            bad += int(orc.lib.orc_select_check_boundaries(i.h, per_side, ctypes.byref(c)))
            checked += int(c.value)
            bad += int(orc.lib.orc_select_check(i.h, 200_000, 17))
            rnd += 200_000
    assert bad == 0, f"oracle select_fast differs from the bisection on {bad} queries"
    log(f"oracle select: {checked} queries around superblock boundaries + {rnd} random ones, directory form == bisection")
    return {"oracle_select_vs_bisection": {"boundary_queries": checked, "random_queries": rnd, "mismatches": 0}}


def part_traversal(inp, res, orc, po, idx):
    """The oracle's own traversal against the GPU's: counters and every bit vector (SHA-256 + word for word)."""
    with Heartbeat(f"oracle traversal (phases 2-3), {THREADS} threads") as hb_t:
        of = _oracle.OracleFlags(orc, idx[0], idx[1] if inp.mode == 2 else None, po.K, po.k_right, threads=THREADS)
    if inp.mode == 3:
        with Heartbeat("oracle document array"):
            d = _oracle._u8(inp.da_host)
            orc.lib.orc_flags_set_da(of.h, d.ctypes.data, len(d))
            of.has_da = True
    oc = of.counters()
    log(f"oracle: {oc}")
    keys = ("leaves", "nodes", "minima", "lcp_values") + (("da_values",) if inp.mode == 2 else ())
    assert all(res.counters[k] == oc[k] for k in keys), ("traversal counters differ", res.counters, oc)
    with Heartbeat("bit vectors: SHA-256, GPU against oracle"):
        want = of.export()
        want_sha = dict(zip(fs.VECTORS, fs.sha256_many(list(want))))
        got_sha, got = res.vector_shas()
        for name, a, b in zip(fs.VECTORS, got, want):
            assert (a is None) == (b is None) and (a is None or np.array_equal(a, b)), f"{name} differs"
    assert got_sha == want_sha
    log(f"flag vectors identical: {want_sha}")
    return of, {"oracle_vector_sha256": want_sha, "gpu_vector_sha256": got_sha, "counters": {k: int(oc[k]) for k in keys},
                "oracle_traversal_seconds": round(hb_t.seconds, 1)}


def neutral_cut(vec, n):
    """A position c near n/2 that no cluster straddles (flag[c-1] == 0, flag = thrK & ~minima): the oracle gives a cluster to the
    piece that holds its BEGIN, the product to the piece that holds its closing position -- at such a c both agree, so the
    two halves of a range cut there can be compared piece by piece."""
    c = n // 2
    flag = vec[0] & ~vec[2]
    while c < n and (int(flag[(c - 1) >> 6]) >> ((c - 1) & 63)) & 1:
        c += 1
    return int(c)


def part_phase4(inp, res, orc, po, idx, of=None, half=None):
    """The oracle's phase 4 over the WHOLE range (half = None) or over one of its two halves ("a": [0, c), "b": [c, n), c a
    cut no cluster straddles -- for inputs whose whole-range pass outlasts a GPU-box call), sharded over the host threads; on
    the GPU's vectors unless the oracle's own (`of`, from part_traversal in the same process) are at hand."""
    rec = {}
    lo, hi = 0, inp.n
    if of is None:
        got_sha, vec = res.vector_shas()
        rec["phase4_ran_on_gpu_vectors_sha256"] = got_sha
        if half is not None:
            c = neutral_cut(vec, inp.n)
            lo, hi = (0, c) if half == "a" else (c, inp.n)
        with Heartbeat("oracle flags imported from the GPU's vectors"):
            of = _oracle.OracleFlags(orc, idx[0], idx[1] if inp.mode == 2 else None, po.K, po.k_right, vectors=vec)
        del vec
    shards = THREADS * 16
    cuts = [tuple(lo + v for v in ebd.shard_range(hi - lo, i, shards)) for i in range(shards)]
    done = [0]

    def one(be):
        r = of.scan(po, be[0], be[1])
        done[0] += 1
        return r["text"], r["cluster_nr"] - 1, r["events"], r["stats"]
    with Heartbeat(f"oracle phase 4, [{lo}, {hi}), {shards} shards on {THREADS} threads") as hb, ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(one, cuts))
    want = fs.renumber([p[0] for p in parts], [p[1] for p in parts])
    orc_sha = hashlib.sha256(want).hexdigest()
    stats = {k: int(sum(p[3][k] for p in parts)) for k in ("runs_closed", "clusters", "bases_in_runs")}
    log(f"oracle: {want.count(b'>')} records, {sum(p[1] for p in parts)} cluster numbers, sha256 {orc_sha}; {stats}")
    if half is not None:
        # the product on the same half, numbered from 1 like the oracle's piece; and: the whole-run text is the two halves renumbered
        pa = fs.piece(inp, res, 0, lo if half == "b" else hi)
        pb = fs.piece(inp, res, lo if half == "b" else hi, inp.n)
        mine = pa if half == "a" else pb
        assert mine["text"] == want, f"GPU .snp of half {half} differs from the oracle's"
        assert (mine["runs_closed"], mine["clusters"], mine["bases_in_runs"]) == (stats["runs_closed"], stats["clusters"], stats["bases_in_runs"]), "scan counters differ"
        assert fs.renumber([pa["text"], pb["text"]], [pa["cluster_numbers"], pb["cluster_numbers"]]) == res.text, "the whole-run .snp is not its two halves renumbered"
        rec.update({"cut": int(lo if half == "b" else hi), "gpu_snp_sha256": res.snp_sha256, "snp_records": int(res.snp_records), "cluster_numbers": int(res.cluster_nr - 1),
                    f"half_{half}": {"range": [int(lo), int(hi)], "oracle_snp_sha256": orc_sha, "identical_to_gpu": True, "snp_records": int(want.count(b">")),
                                     "cluster_numbers": int(sum(p[1] for p in parts)), "scan_counters": stats, "oracle_phase4_seconds": round(hb.seconds, 1), "oracle_shards": shards}})
        return rec
    st = res.calls.stats
    assert (int(st.runs_closed), int(st.clusters), int(st.bases_in_runs)) == (stats["runs_closed"], stats["clusters"], stats["bases_in_runs"]), "scan counters differ"
    assert res.text == want, "GPU .snp differs from the oracle's whole-range .snp"
    assert res.cluster_nr - 1 == sum(p[1] for p in parts)
    rec.update({"oracle_snp_sha256": orc_sha, "gpu_snp_sha256": res.snp_sha256, "identical": True, "snp_records": int(res.snp_records), "snp_bytes": len(want),
                "cluster_numbers": int(res.cluster_nr - 1), "scan_counters": stats, "oracle_phase4_seconds": round(hb.seconds, 1), "oracle_shards": shards})
    return rec


def gpu_phase4_figures(inp, res):
    """What the product's phase 4 did on this input (the constants of the kernels were tuned on error-free reads: VERDICT r04)."""
    st = res.calls.stats
    rec = {"survivors": int(st.survivors), "chains": int(st.chains), "scan_reruns_first_pass": int(st.scan_reruns), "batch_tiles": int(st.batch_tiles),
           "scan_ms": round(float(st.ms_scan), 3), "extend_ms": round(float(st.ms_extend), 3), "right_ms": round(float(st.ms_right), 3),
           "requests_extend": int(st.requests_extend), "requests_right": int(st.requests_right)}
    b2 = inp.bwts[1] if inp.mode == 2 else None
    again = api.scan_clusters(inp.bwts[0], res.flags, res.params, bwt2=b2).stats      # the size of the batches is remembered: no second run now
    rec["second_pass"] = {"scan_reruns": int(again.scan_reruns), "batch_tiles": int(again.batch_tiles), "scan_ms": round(float(again.ms_scan), 3),
                          "extend_ms": round(float(again.ms_extend), 3), "right_ms": round(float(again.ms_right), 3)}
    if inp.n <= (1 << 38) - 2:
        for b in inp.bwts:
            b.build_psi()
        have, nbytes, voids = inp.bwts[0].psi()
        if have:
            c = api.scan_clusters(inp.bwts[0], res.flags, res.params, bwt2=b2)
            assert hashlib.sha256(c.format_snp(res.params, inp.mode != 1)[0]).hexdigest() == res.snp_sha256, ".snp through the FL table differs"
            rec["fl_table"] = {"void_chunk_fraction": round(voids / max(1, nbytes // 16), 6), "void_chunks": int(voids), "right_ms": round(float(c.stats.ms_right), 3),
                               "requests_right": int(c.stats.requests_right), "snp_identical": True}
    ns = res.flags.stats
    rec["traversal"] = {"seconds": round(float(ns.seconds_traversal), 3), "nodes": int(ns.nodes), "leaves": int(ns.leaves), "launches": int(ns.batches),
                        "rank_queries": int(ns.rank_queries)}
    return rec


def main_config():
    if PART == "merge":
        return merge()
    inp = fs.Inputs(NAME, keep_host=True, log=log)
    res = fs.Result(inp, log=log)
    orc = _oracle.Oracle()
    po = _oracle.Params.from_args([])
    out = {"what": f"BASELINE {fs.CONFIGS[NAME]['baseline']} at full size: the product on one MI355X against the CPU oracle on the box's host threads",
           "input": inp.describe(), "options": "defaults (-L31 -R30 -k16 -g10 -v2 -m3 -c20 -q0)", "oracle_threads": THREADS, "gpu_seconds": res.seconds}
    idx = oracle_indexes(orc, inp)
    inp.hosts = []          # the oracle indexes hold what they need
    of = None
    out.update(select_boundary_check(orc, idx))
    if PART == "select":    # only that: the oracle's select at this size (seconds), written beside the round-3 record
        with open(os.path.join(ROOT, "gpurun_out", f"r04_oracle_select_{NAME}.json"), "w") as f:
            json.dump({k: out[k] for k in ("what", "input", "oracle_threads", "oracle_select_vs_bisection")}, f, indent=1)
        log(json.dumps(out["oracle_select_vs_bisection"]))
        return
    if PART in ("traversal", "all"):
        of, rec = part_traversal(inp, res, orc, po, idx)
        out.update(rec)
    if PART in ("phase4", "all"):
        out.update(part_phase4(inp, res, orc, po, idx, of))
        out["gpu_phase4"] = gpu_phase4_figures(inp, res)
    if PART == "all" and "err_per_million" in fs.CONFIGS[NAME]:
        # the same input under -m 2 (two reads suffice for an allele: sequencing errors that occur twice become clusters) -- same flag
        # vectors (they depend on -k and -R only), another phase 4 on both sides
        args = ["-m", "2"]
        res2 = fs.Result(inp, log=log, args=args, flags=res.flags)
        rec2 = part_phase4(inp, res2, orc, _oracle.Params.from_args(args), idx, of)
        rec2["gpu_phase4"] = gpu_phase4_figures(inp, res2)
        out["m2"] = rec2
    if PART in ("phase4a", "phase4b"):
        out.update(part_phase4(inp, res, orc, po, idx, None, half=PART[-1]))
    with open(out_path(NAME, PART), "w") as f:
        json.dump(out, f, indent=1)
    log(json.dumps(out))


def merge():
    """gpurun_out/r03_fullsize_<name>_{all | traversal + phase4}.json -> profiles/r03_fullsize_<name>.json"""
    if os.path.exists(out_path(NAME, "all")):
        m = json.load(open(out_path(NAME, "all")))
    elif os.path.exists(out_path(NAME, "phase4a")):
        t, pa, pb = (json.load(open(out_path(NAME, k))) for k in ("traversal", "phase4a", "phase4b"))
        assert t["input"] == pa["input"] == pb["input"], "the parts ran on different inputs"
        assert pa["phase4_ran_on_gpu_vectors_sha256"] == pb["phase4_ran_on_gpu_vectors_sha256"] == t["oracle_vector_sha256"], "phase 4 ran on vectors the oracle's traversal did not produce"
        assert pa["cut"] == pb["cut"] and pa["gpu_snp_sha256"] == pb["gpu_snp_sha256"]
        m = dict(t)
        m["gpu_seconds"] = pb["gpu_seconds"]
        m.update({k: pb[k] for k in ("cut", "snp_records", "cluster_numbers")})
        m["half_a"], m["half_b"] = pa["half_a"], pb["half_b"]
        # each half of the product's .snp equals the oracle's half, and the product's whole-run text is its two halves renumbered
        # (both asserted at record time): the product's whole-run SHA-256 IS the oracle's
        m["oracle_snp_sha256"] = pb["gpu_snp_sha256"]
        m["scan_counters"] = {k: pa["half_a"]["scan_counters"][k] + pb["half_b"]["scan_counters"][k] for k in pa["half_a"]["scan_counters"]}
        m["identical"] = True
        m["parts"] = ("traversal, phase 4 of [0, cut) and phase 4 of [cut, n) in three GPU-box calls (each inside the 20-minute cap); no cluster straddles the cut; "
                      "phase 4 ran on the GPU's vectors, SHA-256 equal to the oracle's own")
    else:
        t, p4 = json.load(open(out_path(NAME, "traversal"))), json.load(open(out_path(NAME, "phase4")))
        assert t["input"] == p4["input"], "the two parts ran on different inputs"
        assert p4["phase4_ran_on_gpu_vectors_sha256"] == t["oracle_vector_sha256"], "phase 4 ran on vectors the oracle's traversal did not produce"
        m = dict(t)
        m["gpu_seconds"] = p4["gpu_seconds"]
        for k, v in p4.items():
            if k not in m:
                m[k] = v
        m["parts"] = "traversal and phase 4 in two GPU-box calls (each inside the 20-minute cap); phase 4 ran on the GPU's vectors, SHA-256 equal to the oracle's own"
    with open(fs.profile_path(NAME), "w") as f:
        json.dump(m, f, indent=1)
    print("wrote", fs.profile_path(NAME))


def main_replicated():
    """Mode 1 at sizes towards BASELINE configs[4] on ONE GPU: the bench's sample with every read COPIES times in a row --
    eBWT = base eBWT with every symbol repeated (tests/test_sim.py) -- e.g. 50 M reads x10 = 75.5 G positions, 17
    superblocks, positions beyond 2^36.  GPU against the oracle: all bit vectors at full size, phase 4 on windows around
    EVERY superblock boundary, both ends and random places."""
    import torch
    genome_len = int(READS * READ_LEN / COVERAGE)
    g = sim.random_genome(genome_len, SEED)
    h2 = sim.mutate_haplotype(g, SEED)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, READS, READ_LEN, SEED)
    del g, h2
    with Heartbeat("base eBWT on the GPU"):
        dev = sim.build_ebwt_gpu(text, starts, READ_LEN)
    del text, starts
    base_host = np.empty(dev.n, dtype=np.uint8)
    sim._hip_copy_to_host(base_host, dev.ptr, dev.n)

    class _Bytes:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}
    base = torch.as_tensor(_Bytes(dev.ptr, dev.n), device="cuda:0")
    big = base[:, None].expand(-1, COPIES).reshape(-1)
    torch.cuda.synchronize()
    del base
    dev.free()
    n = int(big.numel())
    log(f"n = {n} = {n / (1 << 32):.2f} superblocks")
    p = api.make_params()
    t0 = time.time()
    bwt = api.DnaBwt(None, device_ptr=big.data_ptr(), n=n)
    del big
    torch.cuda.empty_cache()
    t_index = time.time() - t0
    t0 = time.time()
    flags = api.navigate_one_bwt(bwt, p)
    t_nav = time.time() - t0
    ns = flags.stats
    log(f"GPU: index {t_index:.2f}s ({bwt.device_bytes / 1e9:.1f} GB), phases 2-3 {t_nav:.2f}s (traversals {ns.seconds_traversal:.2f}s): {ns.nodes} nodes, {ns.leaves} leaves, lcp_values {ns.lcp_values}")
    assert ns.lcp_values == n
    t0 = time.time()
    calls = api.scan_clusters(bwt, flags, p)
    st = calls.stats
    log(f"GPU phase 4 over the whole BWT: {time.time() - t0:.2f}s wall, scan {st.ms_scan:.1f} ms, {st.clusters} clusters analysed, {len(calls)} in the call set")
    got_full, _, ev_full = calls.format_snp(p, False)
    full_sha = hashlib.sha256(got_full).hexdigest()
    log(f"GPU: {ev_full} records, sha256 {full_sha}")
    del calls
    # ---- oracle on the host: the replicated text is rebuilt from the base copy (no 75 GB download)
    orc = _oracle.Oracle()
    po = _oracle.Params.from_args([])
    with Heartbeat("replicated text on the host"):
        host = np.repeat(base_host, COPIES)
    del base_host
    with Heartbeat(f"oracle index, {THREADS} threads"):
        idx = orc.index(host, threads=THREADS)
    assert idx.ok
    idx.buf = None
    del host
    with Heartbeat(f"oracle traversal (phases 2-3), {THREADS} threads"):
        of = _oracle.OracleFlags(orc, idx, None, po.K, po.k_right, threads=THREADS)
    oc = of.counters()
    log(f"oracle: {oc}")
    assert (ns.leaves, ns.nodes, ns.minima, ns.lcp_values) == (oc["leaves"], oc["nodes"], oc["minima"], oc["lcp_values"]), "traversal counters differ"
    with Heartbeat("bit vectors, GPU against oracle"):
        want_vec = of.export()[:3]
        for name, k in (("thrK", 0), ("thrR", 1), ("minima", 2)):
            pick = [None, None, None, None]
            a = flags.export()[k]
            assert np.array_equal(a, want_vec[k]), f"{name} differs"
            del a
    del want_vec
    log("flag vectors identical (thrK, thrR, minima)")
    win = 1 << 21
    rng = np.random.Generator(np.random.PCG64(7))
    centres = [k << 32 for k in range(1, int(n >> 32) + 1)] + [win // 2, n - win // 2] + [int(v) for v in rng.integers(win, n - win, size=12)]
    windows = sorted({(max(0, (c - win // 2) // 64 * 64), min(n, (c - win // 2) // 64 * 64 + win)) for c in centres})
    log(f"{len(windows)} windows of {win} positions")
    with Heartbeat("oracle scan of the windows"), ThreadPoolExecutor(min(THREADS, len(windows))) as ex:
        wants = list(ex.map(lambda be: of.scan(po, be[0], be[1], capture=True), windows))
    recs = 0
    for (wb, we), want_w in zip(windows, wants):
        c = api.scan_clusters(bwt, flags, p, begin=wb, end=we)
        cl, chains, left, right = c.arrays()
        assert len(c) == want_w["count"], (wb, we, len(c), want_w["count"])
        assert np.array_equal(cl, want_w["clusters"]) and np.array_equal(chains, want_w["chains"]), (wb, we)
        assert np.array_equal(left, want_w["left"]) and np.array_equal(right, want_w["right"]), (wb, we)
        t, _, _ = c.format_snp(p, False)
        assert t == want_w["text"], (wb, we)
        recs += t.count(b">")
    log(f"windows identical: {sum(w['count'] for w in wants)} clusters with contexts, {recs} records")
    res = {"what": f"mode 1, the bench's sample with every read {COPIES} times in a row: GPU vs the CPU oracle on one MI355X",
           "input": {"reads_base": READS, "copies": COPIES, "read_len": READ_LEN, "seed": SEED, "n": n, "superblocks": n / (1 << 32)},
           "gpu": {"index_seconds": round(t_index, 2), "index_device_bytes": int(bwt.device_bytes), "phases23_wall_seconds": round(t_nav, 2),
                   "traversal_seconds": round(float(ns.seconds_traversal), 3), "scan_ms": round(float(st.ms_scan), 2), "clusters_analysed": int(st.clusters),
                   "snp_records": int(ev_full), "snp_sha256": full_sha},
           "flag_vectors_identical": True, "counters": {k: int(oc[k]) for k in ("leaves", "nodes", "minima", "lcp_values")},
           "phase4_windows": len(windows), "window_positions": win, "window_clusters": int(sum(w["count"] for w in wants)), "window_records": int(recs),
           "windows_identical": True, "oracle_threads": THREADS}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fullsize_check_x{COPIES}_{READS}.json"), "w") as f:
        json.dump(res, f, indent=1)
    log(json.dumps(res))


def main():
    if COPIES:
        return main_replicated()
    return main_config()


if __name__ == "__main__":
    main()
