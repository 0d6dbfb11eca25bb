#!/bin/bash
# Round 4: how many scan workgroups per CU beside the context kernels now that the right-context walk is half as long (FL table).
#   gpurun -- 'bash scripts/overlap_ab2.sh > gpurun_out/overlap_ab2.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --rank-queries 1048576 --steps ${STEPS:-80} --warmup ${WARMUP:-12}"
for k in ${KS:-default never 3 4 5 6 8}; do
  unset EB_SCAN_OVERLAP
  if [ "$k" = never ]; then export EB_SCAN_OVERLAP=0; elif [ "$k" != default ]; then export EB_SCAN_OVERLAP=$k; fi
  timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EB_SCAN_OVERLAP=$k', 'ms_per_step', d['ms_per_step'], 'min/median/max', d['step_spread']['ms_per_step_min'], d['step_spread']['ms_per_step_median'], d['step_spread']['ms_per_step_max'], 'scan', d['phase4_device_ms']['scan'], 'ext', d['phase4_device_ms']['extend'], 'right', d['phase4_device_ms']['right'], 'ctx', d['phase4_device_ms']['context_span'], 'snp', d['snp_verified'] is not None, flush=True)
"
done
