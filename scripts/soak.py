"""Soak: repeated create / navigate / scan / destroy cycles and a long run of pipelined jobs; HBM in use must not creep."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ebwt2indel_amd import sim, api
reads, _, _ = sim.diploid_sample(200_000, 40_000, 100, seed=3)
e = sim.build_ebwt(reads, threads=8)
p = api.make_params()
def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20
want = None
marks = []
for cyc in range(30):
    g = api.DnaBwt(e)
    fl = api.navigate_one_bwt(g, p)
    jobs = [api.scan_clusters_begin(g, fl, p, device_only=(k % 2 == 1), wire=(k % 4 == 3)) for k in range(4)]
    for k, j in enumerate(jobs):
        if k % 2 == 1:
            j.wait(); j.device_wire() if k % 4 == 3 else j.device_blob()
        c = j.finish()
        if k % 2 == 0:
            t = c.format_snp(p, False)[0]
            want = want or t
            assert t == want
    fl.close(); del g
    marks.append(used())
    if cyc % 10 == 9: print(f"cycle {cyc + 1}: {marks[-1]:.0f} MiB in use", flush=True)
assert marks[-1] <= marks[4] + 64, marks
g = api.DnaBwt(e); fl = api.navigate_one_bwt(g, p)
job = api.scan_clusters_begin(g, fl, p)
for i in range(3000):
    nxt = api.scan_clusters_begin(g, fl, p)
    assert len(job.finish()) > 0
    job = nxt
job.finish()
print(f"3000 pipelined jobs ok, {used():.0f} MiB in use", flush=True)
# the same with the flag of a pipelining caller (rings of page-locked blobs and device temporaries reserved with the first call set,
# the last part of small jobs held back): results unchanged, no creep
base = used()
job = api.scan_clusters_begin(g, fl, p, pipelined=True)
for i in range(3000):
    nxt = api.scan_clusters_begin(g, fl, p, pipelined=True)
    c = job.finish()
    if i % 500 == 0:
        assert c.format_snp(p, False)[0] == want
    job = nxt
job.finish()
print(f"3000 pipelined jobs of a pipelining caller ok, {used():.0f} MiB in use ({base:.0f} before)", flush=True)
assert used() <= base + 512
