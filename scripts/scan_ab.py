"""A/B timing of the phase-4 kernels on a generated workload under EB_SCAN_DEBUG variants."""
import sys, time, os
# the work-skipping switches exist only in the profiling build: make -C ebwt2indel_amd/csrc EXPERIMENTS=1
_exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ebwt2indel_amd", "_build_exp", "libebwt2indel.so")
assert os.path.exists(_exp), "build the profiling library first: make -C ebwt2indel_amd/csrc EXPERIMENTS=1"
os.environ["EB_LIB_PATH"] = _exp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2", "3", "7"]
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
t = time.time(); fl = api.navigate_one_bwt(bwt, p); print(f"n={bwt.n} nav {time.time()-t:.3f}s", flush=True)
for v in variants:
    os.environ["EB_SCAN_DEBUG"] = v
    ms = []
    for _ in range(5):
        c = api.scan_clusters(bwt, fl, p); ms.append((c.stats.ms_scan, c.stats.ms_extend, c.stats.ms_right, c.stats.ms_total))
    a = np.array(ms[1:]).mean(axis=0)
    print(f"dbg={v}: scan {a[0]:.3f} ms  extend {a[1]:.3f}  right {a[2]:.3f}  total {a[3]:.3f}  runs={c.stats.runs_closed} clusters={c.stats.clusters} surv={c.stats.survivors}  -> {0.75*bwt.n/a[0]/1e6:.1f} GB/s", flush=True)
