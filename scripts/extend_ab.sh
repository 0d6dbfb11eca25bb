#!/bin/bash
# Round 4: k_extend1 (one lane per chain) against k_extend (a quad per chain), alternating on one box.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --rank-queries 1048576 --steps ${STEPS:-80} --warmup ${WARMUP:-12}"
run() { name=$1; shift
  env "$@" timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['phase4_device_ms']['alone']
print('$name', 'ms_per_step', d['ms_per_step'], 'min/median/max', d['step_spread']['ms_per_step_min'], d['step_spread']['ms_per_step_median'], d['step_spread']['ms_per_step_max'], 'scan', d['phase4_device_ms']['scan'], 'ext', d['phase4_device_ms']['extend'], 'right', d['phase4_device_ms']['right'], 'ctx', d['phase4_device_ms']['context_span'], 'alone: scan', a['scan_ms'], 'ext', a['extend_ms'], 'right', a['right_ms'], 'snp', d['snp_verified'] is not None, flush=True)
"
}
for rep in 1 2; do
run quad EB_EXTEND_LANES=4
run one_lane EB_EXTEND_LANES=1
run one_lane_ov6 EB_EXTEND_LANES=1 EB_SCAN_OVERLAP=6
done
