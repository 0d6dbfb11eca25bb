"""The bench's own input at full size (BASELINE configs[1]: 50 M x 150 bp, n = 7.55 G, seed 1) through the CPU oracle.

The unmodified reference needs ~45 min for this input (single thread, 2.9 M positions/s at DRAM-resident sizes: SURVEY.md
section 6), more than a GPU-box call may last; its word at this scale is taken on the replicated input of
scripts/ref_past_2p32.py.  Here the ORACLE -- pinned to the reference by the golden fixtures, by the fresh-input runs
against the reference binary and, past 2^32, by tests/test_gpu_superblocks.py -- walks the very eBWT the bench builds,
on the GPU box's host threads (threaded index + traversal, scan sharded over threads), and its .snp must equal the GPU's
byte for byte.  The SHA-256 goes to profiles/r02_fullsize_config2.json; bench.py asserts it on every default run.

    gpurun --timeout 1200 -- 'python scripts/fullsize_check.py > gpurun_out/fullsize.log 2>&1'
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from ebwt2indel_amd import api, sim  # noqa: E402
from ebwt2indel_amd import distributed as ebd  # noqa: E402
from tests import _oracle  # noqa: E402

READS = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
MODE3 = len(sys.argv) > 2 and sys.argv[2] == "mode3"   # READS reads per sample, two samples: merged BWT + document array
MODE2 = len(sys.argv) > 2 and sys.argv[2] == "mode2"   # the same two samples as two BWTs
COPIES = int(sys.argv[2][1:]) if len(sys.argv) > 2 and sys.argv[2].startswith("x") else 0   # "x10": every read 10 times (sim.replicated_ebwt_gpu)
READ_LEN, COVERAGE, SEED = 150, 30.0, 1
THREADS = int(os.environ.get("EB_CHECK_THREADS", str(min(os.cpu_count() or 8, 32))))


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
        log(f"{self.what}: {time.time() - self.t0:.1f}s")


def main_mode3():
    """BASELINE configs[3] when READS = 50 M: two samples of READS reads each (one per haplotype, scripts/modes23_scale.py's
    recipe), merged eBWT + document array, mode 3 -- GPU against the oracle (threaded index + traversal, scan sharded)."""
    n_each = READS
    glen = int(2 * n_each * READ_LEN / 30)
    g = sim.random_genome(glen, 5)
    h2 = sim.mutate_haplotype(g, 5)
    text, offs = sim.haplotype_text([g, h2])
    rng = np.random.Generator(np.random.PCG64(99))
    s0 = (rng.integers(0, len(g) - READ_LEN + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
    s1 = (rng.integers(0, len(h2) - READ_LEN + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
    del g, h2
    with Heartbeat("merged eBWT + DA on the GPU"):
        dev = sim.build_ebwt_gpu(text, np.concatenate([s0, s1]), READ_LEN, n_first=n_each, want_da=True)
    del text, s0, s1
    n = dev.n
    log(f"n = {n}")
    host, da_host = dev.to_host()
    p = api.make_params()
    t0 = time.time()
    bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=n)
    dev.free()
    flags = api.navigate_one_bwt(bwt, p)
    flags.set_da(da_host)
    calls = api.scan_clusters(bwt, flags, p)
    got, cluster_nr, events = calls.format_snp(p, True)
    t_gpu = time.time() - t0
    gpu_sha = hashlib.sha256(got).hexdigest()
    log(f"GPU mode 3: {got.count(b'>')} records, {len(calls)} clusters in the call set, sha256 {gpu_sha} ({t_gpu:.2f}s)")
    ns = flags.stats
    orc = _oracle.Oracle()
    po = _oracle.Params.from_args([])
    with Heartbeat(f"oracle index, {THREADS} threads"):
        idx = orc.index(host, threads=THREADS)
    with Heartbeat(f"oracle traversal (phases 2-3), {THREADS} threads"):
        of = _oracle.OracleFlags(orc, idx, None, po.K, po.k_right, threads=THREADS)
    with Heartbeat("oracle document array"):
        d = _oracle._u8(da_host)
        orc.lib.orc_flags_set_da(of.h, d.ctypes.data, len(d))
        of.has_da = True
    oc = of.counters()
    log(f"oracle: {oc}")
    assert (ns.leaves, ns.nodes, ns.minima, ns.lcp_values) == (oc["leaves"], oc["nodes"], oc["minima"], oc["lcp_values"]), "traversal counters differ"
    for name, a, b in zip(("thrK", "thrR", "minima", "da"), flags.export(), of.export()):
        assert np.array_equal(a, b), f"{name} differs"
    log("flag vectors identical (thrK, thrR, minima, da)")
    # Phase 4.  The oracle runs the reference's algorithm, which in the two-sample modes extends the contexts of nearly every
    # cluster (one frequent letter on each side is enough, ebwt2InDel.cpp:1042-1052): at 15 G positions that is hours of CPU
    # even on 32 threads (the first attempt of this script gave up after 13 minutes of its first pass).  Small inputs are
    # compared in full; at scale the comparison is on WINDOWS: around every 2^32-symbol superblock boundary, at both ends of
    # the BWT, and at seeded random places -- records (with contexts) and text of each window, GPU against oracle.
    win = 1 << 22
    if n <= (1 << 28):
        windows = [(0, n)]
    else:
        rng = np.random.Generator(np.random.PCG64(7))
        centres = [k << 32 for k in range(1, int(n >> 32) + 1)] + [win // 2, n - win // 2] + [int(v) for v in rng.integers(win, n - win, size=14)]
        windows = sorted({(max(0, (c - win // 2) // 64 * 64), min(n, (c - win // 2) // 64 * 64 + win)) for c in centres})
    log(f"{len(windows)} windows of {win if len(windows) > 1 else n} positions: {[w[0] for w in windows]}")

    def oracle_window(be):
        return of.scan(po, be[0], be[1], capture=True)
    with Heartbeat("oracle scan of the windows"), ThreadPoolExecutor(min(THREADS, len(windows))) as ex:
        wants = list(ex.map(oracle_window, windows))
    recs = 0
    got_parts, want_parts = [], []
    for (b, e), want_w in zip(windows, wants):
        c = api.scan_clusters(bwt, flags, p, begin=b, end=e)
        cl, chains, left, right = c.arrays()
        assert len(c) == want_w["count"], (b, e, len(c), want_w["count"])
        assert np.array_equal(cl, want_w["clusters"]) and np.array_equal(chains, want_w["chains"]), (b, e)
        assert np.array_equal(left, want_w["left"]) and np.array_equal(right, want_w["right"]), (b, e)
        t, _, _ = c.format_snp(p, True)
        assert t == want_w["text"], (b, e)
        recs += t.count(b">")
        got_parts.append(t)
        want_parts.append(want_w["text"])
    log(f"windows identical: {sum(w['count'] for w in wants)} clusters with contexts, {recs} records")
    want = b"".join(want_parts)
    if len(windows) > 1:
        got_full_sha = gpu_sha
        got = b"".join(got_parts)
        gpu_sha = hashlib.sha256(got).hexdigest()
    orc_sha = hashlib.sha256(want).hexdigest()
    res = {"what": "BASELINE configs[3] scale (mode 3: merged BWT + document array) at full size: GPU .snp vs the CPU oracle's",
           "input": {"reads_per_sample": n_each, "read_len": READ_LEN, "n": int(n), "recipe": "scripts/modes23_scale.py (genome seed 5, starts seed 99)",
                     "ebwt_sha256": hashlib.sha256(host).hexdigest(), "da_sha256": hashlib.sha256(da_host).hexdigest()},
           "options": "defaults", "phase4_compared_on": [list(w) for w in windows], "window_clusters": int(sum(w["count"] for w in wants)),
           "gpu_windows_snp_sha256": gpu_sha, "oracle_windows_snp_sha256": orc_sha, "identical": got == want,
           "gpu_whole_snp_sha256": got_full_sha if len(windows) > 1 else gpu_sha, "window_records": int(got.count(b">")), "flag_vectors_identical": True,
           "counters": {k: int(oc[k]) for k in ("leaves", "nodes", "minima", "lcp_values")}, "oracle_threads": THREADS, "gpu_seconds": round(t_gpu, 2)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fullsize_check_mode3_{READS}.json"), "w") as f:
        json.dump(res, f, indent=1)
    log(json.dumps(res))
    assert got == want, "GPU .snp differs from the oracle's at full size (mode 3)"


def main_mode2():
    """BASELINE configs[2] when READS = 25 M: two samples of READS reads each as TWO BWTs (mode 2): the GPU's dual-BWT
    traversal (DA rebuilt from interval arithmetic) and phase 4 against the oracle (threaded two-BWT traversal; phase 4 on
    windows, see main_mode3)."""
    n_each = READS
    glen = int(2 * n_each * READ_LEN / 30)
    g = sim.random_genome(glen, 5)
    h2 = sim.mutate_haplotype(g, 5)
    text, offs = sim.haplotype_text([g, h2])
    rng = np.random.Generator(np.random.PCG64(99))
    s0 = (rng.integers(0, len(g) - READ_LEN + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
    s1 = (rng.integers(0, len(h2) - READ_LEN + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
    del g, h2
    hosts, gpus = [], []
    t_gpu = 0.0
    for starts in (s0, s1):
        with Heartbeat("eBWT on the GPU"):
            dev = sim.build_ebwt_gpu(text, starts, READ_LEN)
        h, _ = dev.to_host()
        hosts.append(h)
        t0 = time.time()
        gpus.append(api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n))
        t_gpu += time.time() - t0
        dev.free()
    del text, s0, s1
    a, b = gpus
    n = a.n + b.n
    log(f"n = {a.n} + {b.n}")
    p = api.make_params()
    t0 = time.time()
    flags = api.navigate_two_bwts(a, b, p)
    calls = api.scan_clusters(a, flags, p, bwt2=b)
    got_full, _, _ = calls.format_snp(p, True)
    t_gpu += time.time() - t0
    full_sha = hashlib.sha256(got_full).hexdigest()
    log(f"GPU mode 2: {got_full.count(b'>')} records, {len(calls)} clusters in the call set, sha256 {full_sha} ({t_gpu:.2f}s)")
    ns = flags.stats
    orc = _oracle.Oracle()
    po = _oracle.Params.from_args([])
    with Heartbeat(f"oracle indexes, {THREADS} threads"):
        i1, i2 = orc.index(hosts[0], threads=THREADS), orc.index(hosts[1], threads=THREADS)
    with Heartbeat(f"oracle two-BWT traversal (phases 2-3), {THREADS} threads"):
        of = _oracle.OracleFlags(orc, i1, i2, po.K, po.k_right, threads=THREADS)
    oc = of.counters()
    log(f"oracle: {oc}")
    assert (ns.leaves, ns.nodes, ns.minima, ns.lcp_values, ns.da_values) == (oc["leaves"], oc["nodes"], oc["minima"], oc["lcp_values"], oc["da_values"]), "traversal counters differ"
    for name, va, vb in zip(("thrK", "thrR", "minima", "da"), flags.export(), of.export()):
        assert np.array_equal(va, vb), f"{name} differs"
    log("flag vectors identical (thrK, thrR, minima, da)")
    win = 1 << 22
    if n <= (1 << 28):
        windows = [(0, n)]
    else:
        rng = np.random.Generator(np.random.PCG64(7))
        centres = [k << 32 for k in range(1, int(n >> 32) + 1)] + [win // 2, n - win // 2] + [int(v) for v in rng.integers(win, n - win, size=14)]
        windows = sorted({(max(0, (c - win // 2) // 64 * 64), min(n, (c - win // 2) // 64 * 64 + win)) for c in centres})
    log(f"{len(windows)} windows of {win if len(windows) > 1 else n} positions: {[w[0] for w in windows]}")
    with Heartbeat("oracle scan of the windows"), ThreadPoolExecutor(min(THREADS, len(windows))) as ex:
        wants = list(ex.map(lambda be: of.scan(po, be[0], be[1], capture=True), windows))
    recs = 0
    for (wb, we), want_w in zip(windows, wants):
        c = api.scan_clusters(a, flags, p, bwt2=b, begin=wb, end=we)
        cl, chains, left, right = c.arrays()
        assert len(c) == want_w["count"], (wb, we, len(c), want_w["count"])
        assert np.array_equal(cl, want_w["clusters"]) and np.array_equal(chains, want_w["chains"]), (wb, we)
        assert np.array_equal(left, want_w["left"]) and np.array_equal(right, want_w["right"]), (wb, we)
        t, _, _ = c.format_snp(p, True)
        assert t == want_w["text"], (wb, we)
        recs += t.count(b">")
    log(f"windows identical: {sum(w['count'] for w in wants)} clusters with contexts, {recs} records")
    res = {"what": "BASELINE configs[2] scale (mode 2: two BWTs) at full size: GPU vs the CPU oracle",
           "input": {"reads_per_sample": n_each, "read_len": READ_LEN, "n1": int(a.n), "n2": int(b.n), "recipe": "scripts/modes23_scale.py (genome seed 5, starts seed 99)",
                     "ebwt_sha256": [hashlib.sha256(h).hexdigest() for h in hosts]},
           "options": "defaults", "flag_vectors_identical": True, "counters": {k: int(oc[k]) for k in ("leaves", "nodes", "minima", "lcp_values", "da_values")},
           "phase4_compared_on": [list(w) for w in windows], "window_clusters": int(sum(w["count"] for w in wants)), "window_records": int(recs), "identical": True,
           "gpu_whole_snp_sha256": full_sha, "gpu_whole_snp_records": int(got_full.count(b">")), "oracle_threads": THREADS, "gpu_seconds": round(t_gpu, 2)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fullsize_check_mode2_{READS}.json"), "w") as f:
        json.dump(res, f, indent=1)
    log(json.dumps(res))


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
    if MODE3:
        return main_mode3()
    if MODE2:
        return main_mode2()
    genome_len = int(READS * READ_LEN / COVERAGE)
    g = sim.random_genome(genome_len, SEED)
    h2 = sim.mutate_haplotype(g, SEED)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, READS, READ_LEN, SEED)
    del g, h2
    with Heartbeat("eBWT on the GPU"):
        dev = sim.build_ebwt_gpu(text, starts, READ_LEN)
    del text, starts
    n = dev.n
    log(f"n = {n}")
    host = np.empty(n, dtype=np.uint8)
    sim._hip_copy_to_host(host, dev.ptr, n)
    p = api.make_params()
    t0 = time.time()
    bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=n)
    dev.free()
    flags = api.navigate_one_bwt(bwt, p)
    calls = api.scan_clusters(bwt, flags, p)
    got, cluster_nr, events = calls.format_snp(p, False)
    t_gpu = time.time() - t0
    gpu_sha = hashlib.sha256(got).hexdigest()
    log(f"GPU: {events} records, {len(calls)} clusters in the call set, sha256 {gpu_sha} ({t_gpu:.2f}s index + traversal + scan + text)")
    ns = flags.stats

    orc = _oracle.Oracle()
    po = _oracle.Params.from_args([])
    with Heartbeat(f"oracle index, {THREADS} threads"):
        idx = orc.index(host, threads=THREADS)
    assert idx.ok
    with Heartbeat(f"oracle traversal (phases 2-3), {THREADS} threads"):
        of = _oracle.OracleFlags(orc, idx, None, po.K, po.k_right, threads=THREADS)
    oc = of.counters()
    log(f"oracle: {oc}")
    assert (ns.leaves, ns.nodes, ns.minima, ns.lcp_values) == (oc["leaves"], oc["nodes"], oc["minima"], oc["lcp_values"]), "traversal counters differ"
    # flag vectors, bit for bit
    for name, a, b in zip(("thrK", "thrR", "minima"), flags.export()[:3], of.export()[:3]):
        assert np.array_equal(a, b), f"{name} differs"
    log("flag vectors identical (thrK, thrR, minima)")
    # phase 4 sharded over the host threads: pass 1 finds how many cluster numbers each shard uses, pass 2 writes the text
    cuts = [ebd.shard_range(n, i, THREADS * 4) for i in range(THREADS * 4)]
    with Heartbeat("oracle scan, pass 1"), ThreadPoolExecutor(THREADS) as ex:
        used = list(ex.map(lambda be: of.scan(po, be[0], be[1])["cluster_nr"] - 1, cuts))
    first = np.concatenate([[1], 1 + np.cumsum(used)[:-1]]).astype(int)
    with Heartbeat("oracle scan, pass 2"), ThreadPoolExecutor(THREADS) as ex:
        parts = list(ex.map(lambda a: of.scan(po, a[0][0], a[0][1], cluster_nr=int(a[1]))["text"], zip(cuts, first)))
    want = b"".join(parts)
    orc_sha = hashlib.sha256(want).hexdigest()
    log(f"oracle: {want.count(b'>')} records, sha256 {orc_sha}")
    res = {"what": "BASELINE configs[1] at full size, bench.py's own input: GPU .snp vs the CPU oracle's (threaded index + traversal, scan sharded over host threads)",
           "input": {"reads": READS, "read_len": READ_LEN, "coverage": COVERAGE, "seed": SEED, "n": int(n), "ebwt_sha256": hashlib.sha256(host).hexdigest()},
           "options": "defaults (-L31 -R30 -k16 -g10 -v2 -m3 -c20 -q0)",
           "gpu_snp_sha256": gpu_sha, "oracle_snp_sha256": orc_sha, "identical": got == want, "snp_records": int(events), "snp_bytes": len(got),
           "flag_vectors_identical": True, "counters": {k: int(oc[k]) for k in ("leaves", "nodes", "minima", "lcp_values")}, "oracle_threads": THREADS,
           "why_not_the_reference_binary": "single-threaded, ~45 min for 7.55 G positions; a GPU-box call is capped at 20 min. It vouches past 2^32 "
                                           "on the replicated input of profiles/r02_reference_past_2p32.json instead."}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"fullsize_check_{READS}.json"), "w") as f:
        json.dump(res, f, indent=1)
    log(json.dumps(res))
    assert got == want, "GPU .snp differs from the oracle's at full size"


if __name__ == "__main__":
    main()
