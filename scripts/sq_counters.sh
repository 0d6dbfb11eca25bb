#!/bin/bash
# What bounds k_scan with the chip to itself: SQ counters of a short, unpipelined bench run, one small counter group per pass
# (rocprofv3 --kernel-trace --pmc only).   gpurun -- 'bash scripts/sq_counters.sh'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --no-pipeline --steps 3 --warmup 1 --rank-queries 1048576"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_LEVEL_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/g$i" -o g$i --output-format csv -- $B > "$OUT/g$i.json" 2> "$OUT/g$i.err" || echo "group $i ($grp) failed" >> "$OUT/failed.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            for key in ("k_scan<1", "k_extend", "k_right<false, 1", "k_right<false, 4", "k_nodes1c", "k_nodes1p", "k_leaves1c", "k_leaves<1"):
                if key in name:
                    acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(out, "sq_summary.txt"), "w") as f:
    for k in sorted(acc):
        f.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            f.write(f"  {c:28s} mean per launch {sum(v)/len(v):.6g}  (launches {len(v)})\n")
print(open(os.path.join(out, "sq_summary.txt")).read())
PY
find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete; find "$OUT" -name '*.db' -delete
du -sh "$OUT"
