"""Durations of the ordering step's kernels (run under rocprofv3 --kernel-trace): phase 4 on a share of the bench's input, one job at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
from ebwt2indel_amd import distributed as ebd
n_reads = 50_000_000; L = 150
share = int(sys.argv[1]) if len(sys.argv) > 1 else 8
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p)
b, e = ebd.shard_range(bwt.n, share // 2, share)
mode = os.environ.get("OP_MODE", "plain")
if "psi" in mode:
    bwt.build_psi()
if "dist" in mode:
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
    t = torch.zeros(1, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
if "five" in mode:                       # five jobs in flight, as bench.py's pipelined steps keep them
    from collections import deque
    q = deque(api.scan_clusters_begin(bwt, fl, p, begin=b, end=e, device_only=True, wire=True, pipelined=True) for _ in range(5))
    for k in range(30):
        j = q.popleft(); j.wait(); j.device_wire(); st = j.finish().stats
        q.append(api.scan_clusters_begin(bwt, fl, p, begin=b, end=e, device_only=True, wire=True, pipelined=True))
    while q:
        st = q.popleft().finish().stats
for k in range(0 if "five" in mode else 12):
    if "wire" in mode or "pipe" in mode:
        job = api.scan_clusters_begin(bwt, fl, p, begin=b, end=e, device_only="wire" in mode, wire="wire" in mode, pipelined="pipe" in mode)
        if "wire" in mode:
            job.wait(); job.device_wire()
        st = job.finish().stats
    else:
        st = api.scan_clusters(bwt, fl, p, begin=b, end=e).stats
print(f"[{mode}] share 1/{share}: survivors {st.survivors} scan {st.ms_scan:.3f} extend {st.ms_extend:.3f} right {st.ms_right:.3f} total {st.ms_total:.3f}", flush=True)
