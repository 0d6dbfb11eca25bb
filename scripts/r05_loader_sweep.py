"""Round 5: the loader's reader threads / piece size on the 7.55 GB file (page cache), through the executable."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, _lib
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
ebwt, _ = dev.to_host(); dev.free()
path = "/tmp/scale.ebwt"; ebwt.tofile(path)
del ebwt, text, starts
for rep in range(2):
    for piece in (32, 64, 16):
        for th in (8, 12, 16, 24):
            env = dict(os.environ, EB_TRACE="index", EB_INDEX_PINNED=f"1,{piece << 20},{th}")
            t = time.time()
            r = subprocess.run([_lib.CLI_PATH, "-1", path, "-o", "/tmp/scale.snp"], capture_output=True, text=True, env=env)
            dt = time.time() - t
            tl = r.stdout.splitlines()[-1] if r.stdout else ""
            idx = [l for l in r.stderr.splitlines() if l.startswith("[index]")]
            print(f"piece {piece} MB x {th} threads: {dt:.2f}s | {tl[:32]} | {idx[0][20:] if idx else ''}", flush=True)
os.remove(path)
