"""Where the executable's start-up goes: the whole run on a tiny input (nothing but start-up and fixed costs), and the dynamic loader's share."""
import os, subprocess, sys, time, gzip, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ebwt2indel_amd import _lib
src = os.path.join(ROOT, "tests", "golden", "dip.ebwt.gz")
tmp = "/tmp/dip.ebwt"
open(tmp, "wb").write(gzip.open(src).read())
for label, env in (("plain", {}), ("LD_BIND_NOW", {"LD_BIND_NOW": "1"})):
    ts = []
    for _ in range(6):
        t = time.time(); r = subprocess.run([_lib.CLI_PATH, "-1", tmp, "-o", "/tmp/dip.snp"], capture_output=True, text=True, env=dict(os.environ, **env)); ts.append(time.time() - t)
    print(label, "wall s:", " ".join(f"{x:.3f}" for x in ts), "|", [l for l in r.stdout.splitlines() if l.startswith("Timings")][0][:160])
r = subprocess.run([_lib.CLI_PATH, "-1", tmp, "-o", "/tmp/dip.snp"], capture_output=True, text=True, env=dict(os.environ, LD_DEBUG="statistics"))
print("\n".join(l for l in r.stderr.splitlines() if "total startup time" in l or "time needed for relocation" in l or "time needed to load objects" in l))
t = time.time(); r = subprocess.run([os.path.join(os.path.dirname(_lib.CLI_PATH), "filter_snp")], capture_output=True); print(f"filter_snp (no HIP, no RCCL): {time.time() - t:.3f} s")
