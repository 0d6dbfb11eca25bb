"""Timeline of the phase-4 kernels of the last few steps from a rocprofv3 --kernel-trace CSV.

    python scripts/step_timeline.py <kernel_trace.csv> [steps]

Prints, for the last `steps` k_scan launches, every kernel dispatch from the first of them on: start offset (us), duration
(us), gap to the previous dispatch's end on the device (us), short name.  The trace is large (the traversal's launches, the
generator's); this keeps the part that shows how a step's kernels pack.
"""
import csv
import re
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
with open(path, newline="") as fh:
    rd = csv.DictReader(fh)
    for r in rd:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
scans = [i for i, r in enumerate(rows) if "k_scan<" in r[2]]
if not scans:
    sys.exit("no k_scan dispatch in the trace")
first = scans[max(0, len(scans) - steps - 1)]
last = scans[-1]
t0 = rows[first][0]
prev_end = t0
busy = 0
print(f"# {len(scans)} k_scan launches in the trace; showing from launch {max(0, len(scans) - steps - 1)} to the last one")
for s, e, name, q in rows[first:last]:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)
    short = re.sub(r"^void ", "", short)
    if "rocprim" in short:
        m = re.search(r"wrapped_(\w+?)_config", name)
        short = "rocprim:" + (m.group(1) if m else "?")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q{q:>3}  {short[:70]}")
    busy += e - s
    prev_end = max(prev_end, e)
span = rows[last][0] - t0
n_steps = len(scans) - 1 - max(0, len(scans) - steps - 1)
print(f"# span {span / 1e3:.1f} us over {n_steps} steps = {span / 1e3 / max(1, n_steps):.1f} us per step; sum of kernel durations {busy / 1e3:.1f} us")
