"""Probe: time eb_index_create_from_file on a generated .ebwt under loader settings (EB_INDEX_COPY_THREADS, EB_INDEX_PINNED_PIECE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu(text, starts, L)
ebwt, _ = dev.to_host(); dev.free()
path = "/tmp/scale.ebwt"; ebwt.tofile(path); del ebwt
for th, piece in [(8, 64), (16, 64), (16, 128), (32, 128), (4, 64), (16, 32)]:
    os.environ["EB_INDEX_COPY_THREADS"] = str(th); os.environ["EB_INDEX_PINNED_PIECE"] = str(piece << 20)
    ts = []
    for _ in range(3):
        t = time.time(); b = api.DnaBwt(path); ts.append(time.time() - t); del b
    print(f"threads={th} piece={piece}MB: {min(ts):.3f}s  ({os.path.getsize(path)/min(ts)/1e9:.1f} GB/s)", flush=True)
os.remove(path)
