"""Round-5 diagnostics on the round-4 tree: where the loader's 0.53 s go, what -G 1 waits for, and the same-box baseline of a rank's eighth."""
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


def run(extra_args, env_extra, label):
    env = dict(os.environ, **env_extra)
    t = time.time()
    r = subprocess.run([_lib.CLI_PATH, "-1", path, "-o", "/tmp/scale.snp"] + extra_args, capture_output=True, text=True, env=env)
    dt = time.time() - t
    print(f"== {label}: {dt:.2f}s rc={r.returncode}", flush=True)
    print("   " + (r.stdout.splitlines()[-1] if r.stdout else ""), flush=True)
    for l in r.stderr.splitlines():
        if l.startswith("["):
            print("   " + l, flush=True)


run([], {"EB_TRACE": "index"}, "plain, defaults (warm-up of the page cache)")
run([], {"EB_TRACE": "index"}, "plain, defaults")
for th in (2, 8, 12, 16):
    run([], {"EB_TRACE": "index", "EB_INDEX_COPY_THREADS": str(th)}, f"plain, {th} copy threads")
for piece in (16, 32, 128, 256):
    run([], {"EB_TRACE": "index", "EB_INDEX_PINNED_PIECE": str(piece << 20)}, f"plain, pieces of {piece} MB")
run([], {"EB_TRACE": "index", "EB_INDEX_PINNED_PIECE": str(32 << 20), "EB_INDEX_COPY_THREADS": "8"}, "plain, 32 MB x 8 threads")
run(["-G", "1"], {"EB_TRACE": "index,multi"}, "-G 1 (RCCL)")
run(["-G", "1"], {"EB_TRACE": "index,multi", "EB_MULTI_BACKEND": "host"}, "-G 1 (host backend: no communicator)")
os.remove(path)
