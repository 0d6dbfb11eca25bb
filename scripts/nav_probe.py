"""Probe: phases 2-3 only, at a given scale, with EB_TRACE=nav output."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    t = time.time(); fl = api.navigate_one_bwt(bwt, p); dt = time.time() - t
    s = fl.stats
    print(f"n={bwt.n} nav {dt:.3f}s nodes={s.nodes} launches={s.batches} -> {s.rank_queries/dt/1e9:.2f} GQ/s (leaves {s.seconds_leaves:.2f}s nodes {s.seconds_nodes:.2f}s)", flush=True)
    fl.close()
