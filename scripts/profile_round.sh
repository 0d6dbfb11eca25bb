#!/bin/bash
# Profiles of one round, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r01'
# Writes under gpurun_out/prof_<tag>/; scripts/pmc_summary.py turns the counter CSVs into profiles/<tag>_pmc_traffic.json.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --cpu-sample-reads 0"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/np" -o np --output-format csv -- $B --no-pipeline > "$OUT/np.json" 2> "$OUT/np.err"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/pl" -o pl --output-format csv -- $B > "$OUT/pl.json" 2> "$OUT/pl.err"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- $B --no-pipeline --steps 5 --warmup 1 > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- $B --no-pipeline --steps 5 --warmup 1 > "$OUT/write.json" 2> "$OUT/write.err"
find "$OUT" -name '*kernel_trace.csv' -delete   # per-dispatch traces are large; the stats and counter tables are what is kept
find "$OUT" -name '*.csv' | head -20
