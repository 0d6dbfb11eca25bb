"""A/B timing of phases 2-3 under EB_NAV_DEBUG variants (profiling build: make -C ebwt2indel_amd/csrc EXPERIMENTS=1;
results of variants other than 0 are wrong by construction -- they only show what each part of k_nodes costs)."""
import sys, time, os
_exp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ebwt2indel_amd", "_build_exp", "libebwt2indel.so")
assert os.path.exists(_exp), "build the profiling library first: make -C ebwt2indel_amd/csrc EXPERIMENTS=1"
os.environ["EB_LIB_PATH"] = _exp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2", "3"]
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p); fl.close()
for v in variants:
    os.environ["EB_NAV_DEBUG"] = v
    for rep in range(2):
        t = time.time(); fl = api.navigate_one_bwt(bwt, p); dt = time.time() - t
        s = fl.stats
        fl.close()
    print(f"dbg={v}: nav {dt:.3f}s nodes={s.nodes} launches={s.batches} (leaves {s.seconds_leaves:.3f}s nodes {s.seconds_nodes:.3f}s)", flush=True)
