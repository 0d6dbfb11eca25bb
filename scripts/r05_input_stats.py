"""What the tuned constants see on an input: records through the traversal's stack (= chains), nodes per chain, and -- with the
profiling build, EB_LIB_PATH=ebwt2indel_amd/_build_exp/libebwt2indel.so -- how many clusters reach k_scan's hot / todo queues.
argv: reads [error_rate_ppm [rc]]      (EB_TRACE=nav,scan is set here; the figures are parsed from the library's trace lines)"""
import os, re, subprocess, sys
if os.environ.get("_INNER") != "1":
    env = dict(os.environ, _INNER="1", EB_TRACE="nav,scan")
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    recs = [int(m.group(1)) for m in re.finditer(r"\[nav\] group \d+: launches=\d+ records=(\d+) pending=0 ", r.stderr)]
    print(r.stdout.strip())
    print("   records through the two stacks at the end of each traversal (leaves, nodes in the order they ended):", recs[-2:] if len(recs) >= 2 else recs)
    for l in r.stderr.splitlines():
        if "to the hot queue" in l:
            print("   " + l)
    sys.exit(r.returncode)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api
reads = int(sys.argv[1]); err = int(sys.argv[2]) if len(sys.argv) > 2 else 0; rc = len(sys.argv) > 3
L = 150
g = sim.random_genome(int(reads * (2 if rc else 1) * L / 30), 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, reads, L, 1)
if err or rc:
    d_reads, d_starts = sim.error_sample_gpu(text, starts, L, 1, err, rc=rc)
    dev = sim.build_ebwt_gpu_slabbed(int(d_reads.data_ptr()), int(d_starts.data_ptr()), L, n_reads=int(d_starts.numel()), text_len=int(d_reads.numel()))
else:
    dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
bwt = api.DnaBwt(None, device_ptr=dev.ptr, n=dev.n); dev.free()
p = api.make_params()
fl = api.navigate_one_bwt(bwt, p); s = fl.stats
c = api.scan_clusters(bwt, fl, p); st = c.stats
print(f"reads={reads} errors={err} ppm rc={rc}: n={bwt.n}; traversal {s.seconds_traversal:.3f} s, {s.nodes} nodes, {s.leaves} leaves, {s.minima} minima; "
      f"phase 4: {st.runs_closed} runs closed, {st.clusters} clusters, {st.survivors} survivors, {len(c)} in the call set, batches of {st.batch_tiles} tiles, {st.scan_reruns} reruns")
