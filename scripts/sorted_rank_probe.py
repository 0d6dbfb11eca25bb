"""Probe: what would a traversal in SA order buy?  Rank queries for 16 M synthetic nodes (six boundaries within ~30 rows
each, like deep suffix-tree nodes) on the config-2 index, nodes in random order vs sorted by position."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ebwt2indel_amd import sim, api
n_reads = int(sys.argv[1]); L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
n = bwt.n
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16 << 20
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
b0 = torch.randint(0, n - 64, (m,), dtype=torch.int64, device="cuda", generator=gen)
off = torch.tensor([0, 5, 11, 17, 23, 30], dtype=torch.int64, device="cuda")
out = torch.empty((m * 6, 4), dtype=torch.int64, device="cuda")
for name, base in (("random order", b0), ("sorted by position", torch.sort(b0).values)):
    pos = (base[:, None] + off[None, :]).reshape(-1).contiguous()
    torch.cuda.synchronize()
    ms = [bwt.rank_device(pos.data_ptr(), m * 6, out.data_ptr()) for _ in range(4)][1:]
    t = sum(ms) / len(ms)
    print(f"{name}: {m} nodes x 6 rank queries in {t:.2f} ms -> {m / t / 1e6:.2f} G nodes/s, {6 * m / t / 1e6:.1f} GQ/s", flush=True)
