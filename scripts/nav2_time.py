"""Phases 2-3 of two BWTs (BASELINE configs[2]: 2 x 25 M x 150 bp) with whatever library EB_LIB_PATH names: traversal seconds, three repetitions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_each = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
L = 150
glen = int(2 * n_each * L / 30)
g = sim.random_genome(glen, 5); h2 = sim.mutate_haplotype(g, 5)
text, offs = sim.haplotype_text([g, h2])
rng = np.random.Generator(np.random.PCG64(99))
s0 = (rng.integers(0, len(g) - L + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
s1 = (rng.integers(0, len(h2) - L + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
p = api.make_params()
a_dev = sim.build_ebwt_gpu_slabbed(text, s0, L); a = api.DnaBwt(None, device_ptr=a_dev.ptr, n=a_dev.n); a_dev.free()
b_dev = sim.build_ebwt_gpu_slabbed(text, s1, L); b = api.DnaBwt(None, device_ptr=b_dev.ptr, n=b_dev.n); b_dev.free()
out = []
for rep in range(4):
    fl = api.navigate_two_bwts(a, b, p); s = fl.stats
    out.append(f"{s.seconds_traversal:.3f} (leaves {s.seconds_leaves:.3f} nodes {s.seconds_nodes:.3f})")
    ok = s.lcp_values == a.n + b.n and s.da_values == a.n + b.n
    fl.close()
print(os.environ.get("EB_LIB_PATH", "default library"), "two-BWT traversal s:", "; ".join(out[1:]), f"nodes={s.nodes} ok={ok}", flush=True)
