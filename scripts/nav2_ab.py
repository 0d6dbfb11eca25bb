"""Two-BWT traversal only (no phase 4) under EB_NAV_DEBUG variants of the profiling build (results other than variant 0 are
wrong by construction): what do the scattered document-array / minima atomics cost?
    make -C ebwt2indel_amd/csrc EXPERIMENTS=1;  python scripts/nav2_ab.py 25000000 0,8,2,10"""
import os, sys, time
_exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ebwt2indel_amd", "_build_exp", "libebwt2indel.so")
assert os.path.exists(_exp), "build the profiling library first: make -C ebwt2indel_amd/csrc EXPERIMENTS=1"
os.environ["EB_LIB_PATH"] = _exp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_each = int(sys.argv[1]); L = 150
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "8", "2", "10"]
glen = int(2 * n_each * L / 30)            # the recipe of scripts/modes23_scale.py
g = sim.random_genome(glen, 5); h2 = sim.mutate_haplotype(g, 5)
text, offs = sim.haplotype_text([g, h2])
rng = np.random.Generator(np.random.PCG64(99))
s0 = (rng.integers(0, len(g) - L + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
s1 = (rng.integers(0, len(h2) - L + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
p = api.make_params()
a_dev = sim.build_ebwt_gpu(text, s0, L); a = api.DnaBwt(None, device_ptr=a_dev.ptr, n=a_dev.n); a_dev.free()
b_dev = sim.build_ebwt_gpu(text, s1, L); b = api.DnaBwt(None, device_ptr=b_dev.ptr, n=b_dev.n); b_dev.free()
fl = api.navigate_two_bwts(a, b, p); fl.close()
for v in variants:
    os.environ["EB_NAV_DEBUG"] = v
    out = []
    for rep in range(3):
        fl = api.navigate_two_bwts(a, b, p); s = fl.stats
        out.append(f"{s.seconds_traversal:.3f}")
        fl.close()                           # never handed to phase 4: with variant 8 the document array is short of ones
    print(f"dbg={v}: traversal s {' '.join(out[1:])} (leaves {s.seconds_leaves:.3f} nodes {s.seconds_nodes:.3f}) nodes={s.nodes} da_values={s.da_values}", flush=True)
