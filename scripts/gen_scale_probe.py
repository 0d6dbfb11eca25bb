"""Probe: time the GPU eBWT generator + phases 1-4 at a given scale (reads x 150 bp, 30x diploid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api

n_reads = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
glen = int(n_reads * L / 30)
t = time.time(); g = sim.random_genome(glen, 1); t1 = time.time()
h2 = sim.mutate_haplotype(g, 1); t2 = time.time()
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1); t3 = time.time()
print(f"host: genome {t1-t:.1f}s mutate {t2-t1:.1f}s text+starts {t3-t2:.1f}s", flush=True)
dev = sim.build_ebwt_gpu(text, starts, L); t4 = time.time()
print(f"gpu generator: n={dev.n} total {t4-t3:.1f}s stages(sa,lcp,keys,sort+emit)={[round(x,2) for x in dev.timings]}", flush=True)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); t5 = time.time()
dev.free()
print(f"index: {t5-t4:.2f}s F={bwt.F_arr}", flush=True)
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p); t6 = time.time()
s = fl.stats
print(f"phases 2-3: {t6-t5:.2f}s nodes={s.nodes} leaves={s.leaves} ranks={s.rank_queries} launches={s.batches} lcp_ok={s.lcp_values==dev.n} -> {s.rank_queries/(t6-t5)/1e9:.2f} GQ/s", flush=True)
calls = api.scan_clusters(bwt, fl, p); t7 = time.time()
st = calls.stats
print(f"phase 4: {t7-t6:.3f}s scan={st.ms_scan:.2f}ms ext={st.ms_extend:.2f}ms right={st.ms_right:.2f}ms clusters={st.clusters} survivors={st.survivors}", flush=True)
txt, cn, ev = calls.format_snp(p, False); t8 = time.time()
print(f"writer: {t8-t7:.2f}s records={ev} bytes={len(txt)}", flush=True)
