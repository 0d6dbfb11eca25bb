#!/bin/bash
# Round 5: SQ counters of the traversal kernels (one BWT: MODE=1; two BWTs: MODE=2), each kernel serialized by the profiler.
#   gpurun -- 'MODE=2 bash scripts/r05_sq_nav.sh'      (rocprofv3 --kernel-trace --pmc only; one small counter group per pass)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
MODE=${MODE:-2}
OUT=$ROOT/gpurun_out/sqnav$MODE
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --mode $MODE --cpu-sample-reads 0 --cpu-dram-reads 0 --no-pipeline --no-other-modes --steps 1 --warmup 0 --rank-queries 1048576"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/g$i" -o g$i --output-format csv -- $B > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i ($grp) failed" >> "$OUT/failed.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
dur = collections.defaultdict(float)
for path in glob.glob(os.path.join(out, "g1", "**", "*kernel_trace.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            for key in ("k_nodes2c", "k_leaves2c", "k_nodes1c", "k_leaves1c", "k_nodes2p", "k_nodes1p", "k_leaves<"):
                if key in row.get("Kernel_Name", ""):
                    dur[key] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6
for path in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            for key in ("k_nodes2c", "k_leaves2c", "k_nodes1c", "k_leaves1c", "k_nodes2p", "k_nodes1p", "k_leaves<"):
                if key in name:
                    acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[key][row["Counter_Name"]] += 1
with open(os.path.join(out, "sq_summary.txt"), "w") as f:
    for k in sorted(acc):
        f.write(f"{k}   (sum over the traversal's launches; kernel time under the profiler {dur.get(k, 0):.1f} ms)\n")
        for c in sorted(acc[k]):
            f.write(f"  {c:28s} {acc[k][c]:.6g}  (launches {cnt[k][c]})\n")
print(open(os.path.join(out, "sq_summary.txt")).read())
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*.db' -delete
du -sh "$OUT"
