"""How much of the PIPELINED phase-4 step each part of k_scan is worth: the profiling build's work-skipping switches (EB_SCAN_DEBUG: 1 no
length histogram, 2 no cluster analysis, 4 no run loop at all, 8 no queues, 16 no statistics; results are WRONG with any of them) under the
same pipelined loop bench.py times, FL table built.     make -C ebwt2indel_amd/csrc EXPERIMENTS=1;  python scripts/scan_ab_pipelined.py 50000000 0,16,8,24"""
import sys, time, os
_exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ebwt2indel_amd", "_build_exp", "libebwt2indel.so")
assert os.path.exists(_exp), "build the profiling library first: make -C ebwt2indel_amd/csrc EXPERIMENTS=1"
os.environ["EB_LIB_PATH"] = _exp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "16", "8", "24"]
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p)
bwt.build_psi()
for v in variants * 2:
    os.environ["EB_SCAN_DEBUG"] = v
    jobs = []
    for k in range(60):
        jobs.append(api.scan_clusters_begin(bwt, fl, p, pipelined=True))
        if k == 11:
            t0 = time.perf_counter()
        if len(jobs) > 3:
            st = jobs.pop(0).finish().stats
    while jobs:
        st = jobs.pop(0).finish().stats
    dt = (time.perf_counter() - t0) / 48 * 1e3
    print(f"dbg={v}: {dt:.3f} ms per pipelined step; last pass: scan {st.ms_scan:.3f} extend {st.ms_extend:.3f} right {st.ms_right:.3f}; survivors {st.survivors}", flush=True)
