"""Phases 2-3 on the bench's input with whatever library EB_LIB_PATH names: traversal seconds, three repetitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
out = []
for rep in range(4):
    fl = api.navigate_one_bwt(bwt, p); s = fl.stats
    out.append(f"{s.seconds_traversal:.3f} (leaves {s.seconds_leaves:.3f} nodes {s.seconds_nodes:.3f})")
    fl.close()
print(os.environ.get("EB_LIB_PATH", "default library"), "traversal s:", "; ".join(out[1:]), f"nodes={s.nodes} lcp_ok={s.lcp_values == bwt.n}", flush=True)
