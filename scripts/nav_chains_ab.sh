#!/bin/bash
# Round 4: k_nodes1c (chains followed in registers) against k_nodes1p (every node through the stack), alternating on one box.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --rank-queries 1048576 --steps 10 --warmup 3"
for rep in 1 2 3; do
for c in 1 0; do
  EB_NAV_CHAINS=$c EB_TRACE=${TRACE:-} timeout -k 10 200 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['phases23_gpu']
print('EB_NAV_CHAINS=$c', 'traversal s', g['seconds'], 'nodes s', g['seconds_nodes'], 'leaves s', g['seconds_leaves'], 'launches', g['launches'], 'nodes', g['nodes'], 'rank GQ/s', g['rank_gq_per_s'], 'snp', d['snp_verified'] is not None, flush=True)
"
done
done
