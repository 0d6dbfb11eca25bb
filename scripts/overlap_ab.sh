#!/bin/bash
# Round-3 experiment: the scan of pass i+1 beside the context kernels of pass i (EB_SCAN_OVERLAP=k: the scan holds k workgroups
# per CU and does not wait for the previous pass's context kernels).  One bench line per setting.
#   gpurun -- 'bash scripts/overlap_ab.sh > gpurun_out/overlap_ab.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --rank-queries 1048576 --steps ${STEPS:-80} --warmup ${WARMUP:-12}"
for k in ${KS:-0 3 4 5}; do
  if [ "$k" = 0 ]; then unset EB_SCAN_OVERLAP; else export EB_SCAN_OVERLAP=$k; fi
  timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('EB_SCAN_OVERLAP=$k', 'ms_per_step', d['ms_per_step'], 'min/median/max', d['step_spread']['ms_per_step_min'], d['step_spread']['ms_per_step_median'], d['step_spread']['ms_per_step_max'], 'scan', d['phase4_device_ms']['scan'], 'ctx', d['phase4_device_ms']['context_span'], 'snp', d['snp_verified'] is not None, 'windows', d['step_spread'].get('windows'), flush=True)
"
done
