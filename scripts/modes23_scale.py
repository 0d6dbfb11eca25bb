"""Modes 2 and 3 at scale: two clean homozygous samples (no shared allele) must give byte-identical .snp through
the two-BWT path (mode 2) and the merged-BWT + document-array path (mode 3) -- SURVEY.md section 0 / Appendix B."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api
n_each = int(sys.argv[1]); L = 150
glen = int(2 * n_each * L / 30)            # each sample covers its haplotype 15x: 30x of the diploid genome
g = sim.random_genome(glen, 5); h2 = sim.mutate_haplotype(g, 5)
text, offs = sim.haplotype_text([g, h2])
rng = np.random.Generator(np.random.PCG64(99))
s0 = (rng.integers(0, len(g) - L + 1, size=n_each, dtype=np.int64) + offs[0]).astype(np.uint64)
s1 = (rng.integers(0, len(h2) - L + 1, size=n_each, dtype=np.int64) + offs[1]).astype(np.uint64)
p = api.make_params()
t = time.time()
da_dev = sim.build_ebwt_gpu_slabbed(text, np.concatenate([s0, s1]), L, n_first=n_each, want_da=True)
m = api.DnaBwt(None, device_ptr=da_dev.ptr, n=da_dev.n)
da_host = np.empty(da_dev.n, dtype=np.uint8); sim._hip_copy_to_host(da_host, da_dev.da_ptr, da_dev.n); da_dev.free()
t_gen3 = time.time() - t
t = time.time(); fl3 = api.navigate_one_bwt(m, p); fl3.set_da(da_host); t_nav3 = time.time() - t
c3 = api.scan_clusters(m, fl3, p); c3 = api.scan_clusters(m, fl3, p)
snp3 = c3.format_snp(p, True)[0]
print(f"mode 3: n={m.n} gen {t_gen3:.1f}s nav {t_nav3:.2f}s (traversals {fl3.stats.seconds_traversal:.3f}s: leaves {fl3.stats.seconds_leaves:.3f} / nodes {fl3.stats.seconds_nodes:.3f}) phase4 {c3.stats.ms_total:.2f} ms (scan {c3.stats.ms_scan:.2f} extend {c3.stats.ms_extend:.2f} right {c3.stats.ms_right:.2f}; {len(c3)} clusters) records={snp3.count(b'>')} sha={hashlib.sha256(snp3).hexdigest()[:16]}", flush=True)
del c3, fl3, m
t = time.time()
a_dev = sim.build_ebwt_gpu_slabbed(text, s0, L); a = api.DnaBwt(None, device_ptr=a_dev.ptr, n=a_dev.n); a_dev.free()
b_dev = sim.build_ebwt_gpu_slabbed(text, s1, L); b = api.DnaBwt(None, device_ptr=b_dev.ptr, n=b_dev.n); b_dev.free()
t_gen2 = time.time() - t
t = time.time(); fl2 = api.navigate_two_bwts(a, b, p); t_nav2 = time.time() - t
st = fl2.stats
c2 = api.scan_clusters(a, fl2, p, bwt2=b); c2 = api.scan_clusters(a, fl2, p, bwt2=b)
snp2 = c2.format_snp(p, True)[0]
print(f"mode 2: n={a.n}+{b.n} gen {t_gen2:.1f}s nav {t_nav2:.2f}s (traversals {st.seconds_traversal:.3f}s: leaves {st.seconds_leaves:.3f} / nodes {st.seconds_nodes:.3f}, {st.rank_queries} rank queries, {st.leaves} leaves; nodes {st.nodes}, da_ok {st.da_values == a.n + b.n}, lcp_ok {st.lcp_values == a.n + b.n}) "
      f"phase4 {c2.stats.ms_total:.2f} ms (scan {c2.stats.ms_scan:.2f} extend {c2.stats.ms_extend:.2f} right {c2.stats.ms_right:.2f}; {len(c2)} clusters) records={snp2.count(b'>')} sha={hashlib.sha256(snp2).hexdigest()[:16]}", flush=True)
da2 = np.unpackbits(fl2.export()[3].view(np.uint8), bitorder="little")[: len(da_host)]
print("DA rebuilt from the two BWTs equals the generator's:", bool(np.array_equal(da2, (da_host == ord('1')).astype(np.uint8))))
print("mode 2 == mode 3:", snp2 == snp3)
