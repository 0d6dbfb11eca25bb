"""Does the ordered hand-over of the scan spill on the bench's data?  Prints eb_scan_stats.scan_reruns and the batch fill."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p)
for rep in range(3):
    c = api.scan_clusters(bwt, fl, p)
    st = c.stats
    print(f"n={bwt.n} survivors={st.survivors} reruns={st.scan_reruns} scan {st.ms_scan:.3f} ms total {st.ms_total:.3f} ms", flush=True)
# where do the survivors sit?  begins per window of 2^21 positions (one batch of the ordered hand-over)
cl = c.arrays()[0]
b = cl.view(np.uint64).reshape(-1, 4)[:, 0]          # eb_cluster = 32 bytes, begin first
e = cl.view(np.uint64).reshape(-1, 4)[:, 1]
print(f"cluster length: mean {(e - b).mean():.1f} max {(e - b).max()}")
for shift in (19, 21):
    ww = np.bincount((e >> np.uint64(shift)).astype(np.int64))
    print(f"  survivors per 2^{shift} closing positions: mean {ww.mean():.1f} max {ww.max()} at window {int(ww.argmax())}; top5 {np.sort(ww)[-5:]}")
w = np.bincount((b >> np.uint64(21)).astype(np.int64))
print(f"survivors per 2^21 positions: mean {w.mean():.1f} max {w.max()} at window {int(w.argmax())}; windows over 512: {(w > 512).sum()}; top5 {np.sort(w)[-5:]}")
