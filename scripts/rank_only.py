"""k_rank4_batch alone on a config-2-sized index (profiling target for the PMC passes of scripts/profile_round.sh):
random symbols (the kernel's traffic does not depend on them), 2^28 uniformly random queries, a few launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ebwt2indel_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_550_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(7)
lut = torch.tensor(list(b"ACGT#"), dtype=torch.uint8, device=dev)
text = torch.empty(n, dtype=torch.uint8, device=dev)
for a in range(0, n, 1 << 28):
    b = min(n, a + (1 << 28))
    text[a:b] = lut[torch.randint(0, 21, (b - a,), dtype=torch.int64, device=dev, generator=g) // 5]
torch.cuda.synchronize()
bwt = api.DnaBwt(None, device_ptr=text.data_ptr(), n=n)
del text
torch.cuda.empty_cache()
pos = torch.randint(0, n + 1, (m,), dtype=torch.int64, device=dev, generator=g)
out = torch.empty((m, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ms = [bwt.rank_device(pos.data_ptr(), m, out.data_ptr()) for _ in range(reps)]
print(f"n={n} index {bwt.device_bytes/1e9:.2f} GB, {m} queries: " + " ".join(f"{t:.3f}" for t in ms) + f" ms -> {m/min(ms[1:])/1e6:.2f} GQ/s")
