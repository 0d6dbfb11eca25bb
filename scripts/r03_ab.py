"""Round-3 A/B on the bench's input (BASELINE configs[1]): phases 2-3 under the traversal switches (pipelined node kernel,
launches per read-back) and phase 4 with whatever library EB_LIB_PATH names (the scan's prefetch is a compile-time switch:
make -C ebwt2indel_amd/csrc OUT=../_build_v_nopf EXTRA=-DEB_SCAN_PREFETCH=0).  Every variant must reproduce the counters and
the .snp SHA-256 of the full-size oracle check (profiles/r02_fullsize_config2.json).

    python scripts/r03_ab.py [reads] [nav|scan|all]
"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from ebwt2indel_amd import sim, api

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
what = sys.argv[2] if len(sys.argv) > 2 else "all"
L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
lib = os.environ.get("EB_LIB_PATH", "default library")
known = None
try:
    known = json.load(open(os.path.join(ROOT, "profiles", "r02_fullsize_config2.json")))
except OSError:
    pass
print(f"[{lib}] n={bwt.n}", flush=True)
ref = None
fl = None
variants = [("pipelined, 8 launches per read-back", {}), ("pipelined, 1 launch per read-back", {"EB_NAV_GROUP": "1"}),
            ("round-2 node kernel, 8 per read-back", {"EB_NAV_PIPELINED": "0"}), ("round-2 node kernel, 1 per read-back", {"EB_NAV_PIPELINED": "0", "EB_NAV_GROUP": "1"}),
            ("pipelined, 16 per read-back", {"EB_NAV_GROUP": "16"}), ("pipelined, 4 per read-back", {"EB_NAV_GROUP": "4"})]
if what == "scan":
    variants = variants[:1]
for name, env in variants:
    for k in ("EB_NAV_GROUP", "EB_NAV_PIPELINED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = []
    for rep in range(3 if what != "scan" else 1):
        if fl is not None:
            fl.close()
        t = time.time()
        fl = api.navigate_one_bwt(bwt, p); s = fl.stats
        out.append(f"{s.seconds_traversal:.3f} (leaves {s.seconds_leaves:.3f} nodes {s.seconds_nodes:.3f} wall {time.time() - t:.3f})")
    cnt = (s.nodes, s.leaves, s.minima, s.lcp_values, s.rank_queries)
    ones = fl.popcounts()
    if ref is None:
        ref = (cnt, ones)
    assert (cnt, ones) == ref, (name, cnt, ones, ref)
    print(f"nav [{name}]: traversal s: " + "; ".join(out) + f"  launches={s.batches} nodes={s.nodes} lcp_ok={s.lcp_values == bwt.n}", flush=True)
for k in ("EB_NAV_GROUP", "EB_NAV_PIPELINED"):
    os.environ.pop(k, None)
if known and n_reads == known["input"]["reads"]:
    c = known["counters"]
    assert (s.nodes, s.leaves, s.minima) == (c["nodes"], c["leaves"], c["minima"]), "traversal counters differ from the full-size oracle run"
if what in ("scan", "all"):
    ms = []
    for _ in range(8):
        c = api.scan_clusters(bwt, fl, p); ms.append((c.stats.ms_scan, c.stats.ms_extend, c.stats.ms_right, c.stats.ms_total))
    a = np.array(ms[2:])
    st = c.stats
    print(f"scan [{lib}]: k_scan {a[:,0].mean():.4f} ms (min {a[:,0].min():.4f})  extend {a[:,1].mean():.4f}  right {a[:,2].mean():.4f}  total {a[:,3].mean():.4f}  "
          f"runs={st.runs_closed} clusters={st.clusters} surv={st.survivors} reruns={st.scan_reruns} -> {0.75*bwt.n/a[:,0].mean()/1e6:.1f} GB/s algorithmic", flush=True)
    text_, _, ev = c.format_snp(p, False)
    sha = hashlib.sha256(text_).hexdigest()
    ok = known is not None and n_reads == known["input"]["reads"] and sha == known["oracle_snp_sha256"]
    print(f"snp sha256 {sha} records {ev} equals the full-size oracle run: {ok}", flush=True)
    if known is not None and n_reads == known["input"]["reads"]:
        assert ok
