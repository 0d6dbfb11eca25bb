"""BASELINE configs[4] SIZE on one MI355X: 151 G BWT positions (36 superblocks) resident in HBM, whole pipeline.

No eBWT of 1 G distinct reads can be built on one GPU (its text's suffix array alone does not fit), and
no CPU checker fits the box's host-memory cap at this size (scripts/fullsize_check.py stops at 75.5 G).  What CAN be
done: the bench's sample (50 M x 150 bp) with every read COPIES = 20 times in a row -- its eBWT is the base eBWT with
every symbol repeated 20 times -- and checks that need no oracle:
  * index: F = 20 x F(base); parallel_rank / LF at random positions and around every one of the 35 superblock boundaries
    against the closed form  rank_rep(i) = c * rank_base(i div c) + (i mod c) * [base[i div c] == letter],  evaluated with
    the (verified) base index on the same GPU;
  * traversal: every LCP slot written once (lcp_values == n), the suffix tree has exactly the base sample's nodes;
  * phase 4: the whole-range call set equals the concatenation of 8 shards' call sets (the N = 8 ownership rule), text
    and cluster numbering included;
  * HBM in use at each stage.

    gpurun --timeout 900 -- 'python scripts/hbm_resident_151g.py > gpurun_out/hbm151.log 2>&1'
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ebwt2indel_amd import api, sim  # noqa: E402
from ebwt2indel_amd import distributed as ebd  # noqa: E402

READS = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
COPIES = int(sys.argv[2]) if len(sys.argv) > 2 else 20
READ_LEN, COVERAGE, SEED = 150, 30.0, 1


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def hbm():
    free_b, total_b = torch.cuda.mem_get_info(0)
    return f"HBM in use {(total_b - free_b) / 1e9:.1f} of {total_b / 1e9:.1f} GB"


class _Bytes:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 3, "strides": None}


def main():
    genome_len = int(READS * READ_LEN / COVERAGE)
    g = sim.random_genome(genome_len, SEED)
    h2 = sim.mutate_haplotype(g, SEED)
    text, offs = sim.haplotype_text([g, h2])
    starts = sim.sample_read_starts([g, h2], offs, READS, READ_LEN, SEED)
    del g, h2
    dev = sim.build_ebwt_gpu(text, starts, READ_LEN)
    del text, starts
    n0 = dev.n
    base_t = torch.as_tensor(_Bytes(dev.ptr, n0), device="cuda:0")
    base_bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=n0)          # the base index: verified at full size (profiles/r02_fullsize_config2.json)
    c = COPIES
    t0 = time.time()
    big = base_t[:, None].expand(-1, c).reshape(-1)
    torch.cuda.synchronize()
    n = int(big.numel())
    log(f"n = {n} = {c} x {n0} = {n / (1 << 32):.2f} superblocks; replicated text built in {time.time() - t0:.2f}s; {hbm()}")
    t0 = time.time()
    bwt = api.DnaBwt(None, device_ptr=big.data_ptr(), n=n)
    t_index = time.time() - t0
    del big
    torch.cuda.empty_cache()
    log(f"index: {t_index:.2f}s, {bwt.device_bytes / 1e9:.1f} GB; {hbm()}")
    assert bwt.F_arr == [c * f for f in base_bwt.F_arr]

    # ---- primitives against the closed form
    rng = np.random.Generator(np.random.PCG64(3))
    edges = np.concatenate([np.zeros(0, dtype=np.uint64)] + [(np.uint64(k) << np.uint64(32)) + np.arange(-300, 300, dtype=np.int64).astype(np.uint64)
                                                              for k in range(1, int(n >> 32) + 1)])
    pos = np.concatenate([edges, rng.integers(0, n + 1, size=2_000_000, dtype=np.uint64), np.array([0, 1, n - 1, n], dtype=np.uint64)])
    q, r = pos // np.uint64(c), pos % np.uint64(c)
    rb = base_bwt.parallel_rank(q)
    sym = np.empty(len(q), dtype=np.uint8)
    inside = q < np.uint64(n0)
    sym[inside] = base_bwt[q[inside]]
    sym[~inside] = 0
    want = rb * np.uint64(c)
    for k, ch in enumerate(b"ACGT"):
        want[:, k] += r * (sym == ch).astype(np.uint64)
    got = bwt.parallel_rank(pos)
    assert np.array_equal(got, want), "parallel_rank differs from the closed form"
    acc = pos[pos < np.uint64(n)]
    assert np.array_equal(bwt[acc], base_bwt[acc // np.uint64(c)])
    # LF(range) = F + rank at both ends
    l = rng.integers(0, n - 500, size=200_000, dtype=np.uint64)
    lr = np.stack([l, l + rng.integers(0, 500, size=len(l), dtype=np.uint64)], axis=1)
    lf = bwt.LF(lr)
    F = np.array(bwt.F_arr, dtype=np.uint64)

    def closed(p):
        qq, rr = p // np.uint64(c), p % np.uint64(c)
        out = base_bwt.parallel_rank(qq) * np.uint64(c)
        s_ = np.zeros(len(qq), dtype=np.uint8)
        ins = qq < np.uint64(n0)
        s_[ins] = base_bwt[qq[ins]]
        for k, ch in enumerate(b"ACGT"):
            out[:, k] += rr * (s_ == ch).astype(np.uint64)
        return out
    wa, wb = closed(lr[:, 0]), closed(lr[:, 1])
    assert np.array_equal(lf[:, 0::2], F[None, :] + wa) and np.array_equal(lf[:, 1::2], F[None, :] + wb)
    # FL is the inverse of LF: row fl holds letter F(i) and LF of [fl, fl+1) by that letter is [i, i+1)
    rows = rng.integers(int(F[0]), n, size=200_000, dtype=np.uint64)
    fl = bwt.FL(rows)
    letters = bwt[fl]
    back = bwt.LF(np.stack([fl, fl + np.uint64(1)], axis=1), letters)[:, 0]
    assert np.array_equal(back, rows)
    log(f"primitives: {len(pos)} rank queries (600 around each of {int(n >> 32)} superblock boundaries), {len(lr)} LF ranges, {len(rows)} FL/LF round trips: all equal the closed form")
    p = api.make_params()
    base_nodes = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_fullsize_config2.json")) as fh:
            base_nodes = json.load(fh)["counters"]["nodes"] if READS == 50_000_000 else None
    except OSError:
        pass
    del base_bwt, base_t
    dev.free()
    torch.cuda.empty_cache()

    # ---- phases 2-3
    t0 = time.time()
    flags = api.navigate_one_bwt(bwt, p)
    t_nav = time.time() - t0
    ns = flags.stats
    log(f"phases 2-3: wall {t_nav:.2f}s, traversals {ns.seconds_traversal:.2f}s: {ns.nodes} nodes, {ns.leaves} leaves, {ns.minima} minima, lcp_values {ns.lcp_values}; {hbm()}")
    assert ns.lcp_values == n
    if base_nodes is not None:
        assert ns.nodes == base_nodes, (ns.nodes, base_nodes)

    # ---- phase 4: whole range vs 8 shards
    t0 = time.time()
    calls = api.scan_clusters(bwt, flags, p)
    st = calls.stats
    whole, cn_whole, ev_whole = calls.format_snp(p, False)
    log(f"phase 4, whole BWT: {time.time() - t0:.2f}s wall (scan {st.ms_scan:.1f} ms, extend {st.ms_extend:.1f}, right {st.ms_right:.1f}); {st.clusters} clusters analysed, "
        f"{len(calls)} in the call set, {ev_whole} records; {hbm()}")
    del calls
    parts, cn, ev = [], 1, 0
    for rk in range(8):
        b, e = ebd.shard_range(n, rk, 8)
        t, cn, ev = api.scan_clusters(bwt, flags, p, begin=b, end=e).format_snp(p, False, cn, ev)
        parts.append(t)
    assert b"".join(parts) == whole and cn == cn_whole and ev == ev_whole
    sha = hashlib.sha256(whole).hexdigest()
    log(f"8 shards (the N = 8 ownership rule) concatenate to the whole-range .snp: {ev} records, sha256 {sha}")
    res = {"what": f"BASELINE configs[4] size on ONE MI355X: the bench's sample with every read {c} times in a row, whole pipeline HBM-resident",
           "n": n, "superblocks": n / (1 << 32), "index_seconds": round(t_index, 2), "index_device_bytes": int(bwt.device_bytes),
           "phases23_wall_seconds": round(t_nav, 2), "traversal_seconds": round(float(ns.seconds_traversal), 2), "nodes": int(ns.nodes), "leaves": int(ns.leaves),
           "lcp_values_equals_n": True, "nodes_equal_base_sample": base_nodes is not None, "phase4_scan_ms": round(float(st.ms_scan), 1),
           "clusters_analysed": int(st.clusters), "snp_records": int(ev_whole), "snp_sha256": sha, "shards8_equal_whole": True,
           "rank_queries_checked": int(len(pos)), "checks": "closed form from the base index (rank, access, LF), FL/LF round trips, lcp_values == n, 8 shards == whole"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"hbm_resident_x{c}_{READS}.json"), "w") as f:
        json.dump(res, f, indent=1)
    log(json.dumps(res))


if __name__ == "__main__":
    main()
