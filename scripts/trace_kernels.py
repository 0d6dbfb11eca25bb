"""Per-kernel duration table of a rocprofv3 --kernel-trace CSV, for the kernels whose name contains one of the given words: calls, mean, min, max (us)."""
import csv, re, sys
from collections import defaultdict
path, words = sys.argv[1], sys.argv[2:]
acc = defaultdict(list)
for r in csv.DictReader(open(path, newline="")):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n); n = re.sub(r"^void ", "", n)
    if not words or any(w in n for w in words):
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{n[:44]:44s} calls {len(v):4d}  mean {sum(v)/len(v):9.1f}  min {v2[0]:9.1f}  median {v2[len(v2)//2]:9.1f}  max {v2[-1]:9.1f} us   last 6: {' '.join(f'{x:.1f}' for x in v[-6:])}")
