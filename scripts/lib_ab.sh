#!/bin/bash
# A/B of two builds of the library on the driver's workload, alternating on one box: one bench line per run.
#   make -C ebwt2indel_amd/csrc OUT=../_build_v_<name> EXTRA=-D...      (variant builds are git-ignored, they travel with gpurun)
#   gpurun -- 'LIBS="_build_v_nocarry _build" bash scripts/lib_ab.sh > gpurun_out/lib_ab.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --rank-queries 1048576 --steps ${STEPS:-80} --warmup ${WARMUP:-12}"
for rep in $(seq 1 ${REPS:-2}); do
  for lib in ${LIBS:-_build}; do
    EB_LIB_PATH=$ROOT/ebwt2indel_amd/$lib/libebwt2indel.so timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', 'ms_per_step', d['ms_per_step'], 'min/median/max', d['step_spread']['ms_per_step_min'], d['step_spread']['ms_per_step_median'], d['step_spread']['ms_per_step_max'], 'scan', d['phase4_device_ms']['scan'], 'scan alone', d['roofline']['alone']['avg_launch_ms'], 'ctx', d['phase4_device_ms']['context_span'], 'alone: extend', d['phase4_device_ms']['alone']['extend_ms'], 'right', d['phase4_device_ms']['alone']['right_ms'], 'span', d['phase4_device_ms']['alone']['context_span_ms'], 'nav s', d['phases23_gpu']['seconds'], 'leaves', d['phases23_gpu']['seconds_leaves'], 'snp', d['snp_verified'] is not None, flush=True)
" || exit 1
  done
done
