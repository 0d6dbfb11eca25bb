"""Check the ebwt2InDel executable on a large file: GPU-generated eBWT -> /tmp file -> CLI -> .snp, against the in-process path."""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ebwt2indel_amd import sim, api, _lib
n_reads = int(sys.argv[1]); L = 150
glen = int(n_reads * L / 30)
g = sim.random_genome(glen, 1); h2 = sim.mutate_haplotype(g, 1)
text, offs = sim.haplotype_text([g, h2]); starts = sim.sample_read_starts([g, h2], offs, n_reads, L, 1)
dev = sim.build_ebwt_gpu_slabbed(text, starts, L)
ebwt, _ = dev.to_host(); dev.free()
path = "/tmp/scale.ebwt"; ebwt.tofile(path)
p = api.make_params()
t = time.time(); want = api.run_one_dataset(ebwt, p); t_api = time.time() - t
t = time.time()
r = subprocess.run([_lib.CLI_PATH, "-1", path, "-o", "/tmp/scale.snp"], capture_output=True, text=True)
t_cli = time.time() - t
got = open("/tmp/scale.snp", "rb").read()
print(f"n={len(ebwt)} api {t_api:.2f}s cli {t_cli:.2f}s rc={r.returncode} identical={got == want} records={got.count(b'>')} sha={hashlib.sha256(got).hexdigest()[:16]}")
print(r.stdout.splitlines()[-1])
for l in r.stderr.splitlines():
    if l.startswith("["):
        print("   " + l[:400])
# the same file through the multi-GPU mode of the executable with one rank (RCCL communicator, device images, unpack + writer)
t = time.time()
r = subprocess.run([_lib.CLI_PATH, "-1", path, "-o", "/tmp/scale_g1.snp", "-G", "1"], capture_output=True, text=True)
t_g1 = time.time() - t
got1 = open("/tmp/scale_g1.snp", "rb").read()
print(f"-G 1: {t_g1:.2f}s rc={r.returncode} identical={got1 == want}")
print(r.stdout.splitlines()[-1] if r.stdout else r.stderr[-500:])
os.remove("/tmp/scale_g1.snp")
for rep in range(int(os.environ.get("CLI_REPEATS", "0"))):     # run-to-run spread of both forms (HBM just freed by the previous process)
    for extra in ([], ["-G", "1"]):
        time.sleep(float(os.environ.get("CLI_SLEEP", "0")))     # (an idle GPU: the driver has wiped what the previous process released)
        t = time.time()
        r = subprocess.run([_lib.CLI_PATH, "-1", path, "-o", "/tmp/scale_rep.snp"] + extra, capture_output=True, text=True)
        print(f"repeat {rep} {' '.join(extra) or 'plain'}: {time.time() - t:.2f}s rc={r.returncode} :: {r.stdout.splitlines()[-1][:160] if r.stdout else r.stderr[-200:]}")
    os.remove("/tmp/scale_rep.snp")
os.remove(path); os.remove("/tmp/scale.snp")
