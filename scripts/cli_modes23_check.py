"""The ebwt2InDel executable in modes 2 and 3 on large files: two-sample eBWTs built on the GPU -> /tmp files -> CLI -> .snp, against the
in-process path; prints the executable's timing lines (profiles/r05_cli_modes23.txt).  argv[1]: reads per collection."""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ebwt2indel_amd import sim, api, _lib
n_each = int(sys.argv[1]); L = 150
text, s0, s1 = sim.two_sample_text(n_each, L)
p = api.make_params()
tmp = os.environ.get("TMPDIR", "/tmp")


def run(args, label, want):
    out = os.path.join(tmp, "m23.snp")
    for rep in range(2):
        time.sleep(float(os.environ.get("CLI_SLEEP", "0")))      # (an idle GPU: the driver has wiped what the previous process released)
        t = time.time()
        r = subprocess.run([_lib.CLI_PATH] + args + ["-o", out], capture_output=True, text=True, env=dict(os.environ, EB_TRACE=os.environ.get("EB_TRACE", "index")))
        dt = time.time() - t
        got = open(out, "rb").read() if r.returncode == 0 else b""
        print(f"{label} run {rep}: {dt:.2f}s rc={r.returncode} identical={got == want} records={got.count(b'>')} sha={hashlib.sha256(got).hexdigest()[:16]}")
        for l in r.stdout.splitlines():
            if l.startswith(("Timings", "Document array")):
                print("   " + l[:600])
        for l in r.stderr.splitlines():
            if l.startswith("["):
                print("   " + l[:400])
    os.remove(out)


# mode 3: merged BWT + document array
dev = sim.build_ebwt_gpu_slabbed(text, np.concatenate([s0, s1]), L, n_first=n_each, want_da=True)
m, da = dev.to_host(); dev.free()
pm, pd = os.path.join(tmp, "m23_m.ebwt"), os.path.join(tmp, "m23_m.da")
m.tofile(pm); da.tofile(pd)
t = time.time(); want3 = api.run_two_datasets_da(m, da, p); t_api = time.time() - t
print(f"mode 3: n={len(m)} in-process (host buffers) {t_api:.2f}s records={want3.count(b'>')}", flush=True)
del m, da
run(["-1", pm, "-d", pd], "mode 3 CLI", want3)
run(["-1", pm, "-d", pd, "-G", "1"], "mode 3 CLI -G 1", want3)
os.remove(pm); os.remove(pd)
# mode 2: two BWTs
files = []
hosts = []
for k, s in enumerate((s0, s1)):
    dev = sim.build_ebwt_gpu_slabbed(text, s, L)
    h, _ = dev.to_host(); dev.free()
    pth = os.path.join(tmp, f"m23_{k}.ebwt"); h.tofile(pth); files.append(pth); hosts.append(h)
t = time.time(); want2 = api.run_two_datasets(hosts[0], hosts[1], p); t_api = time.time() - t
print(f"mode 2: n={len(hosts[0])}+{len(hosts[1])} in-process (host buffers) {t_api:.2f}s records={want2.count(b'>')}", flush=True)
del hosts
run(["-1", files[0], "-2", files[1]], "mode 2 CLI", want2)
for f in files:
    os.remove(f)
