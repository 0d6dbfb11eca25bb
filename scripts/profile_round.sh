#!/bin/bash
# Profiles of one round, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r02'
# Writes under gpurun_out/prof_<tag>/; scripts/pmc_summary.py turns the counter CSVs into profiles/<tag>_pmc_traffic.json.
# Counter passes are separate runs with --kernel-trace only (no other trace domains), one counter group per pass.
set -e -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --cpu-sample-reads 0 --cpu-dram-reads 0 --no-other-modes"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/np" -o np --output-format csv -- $B --no-pipeline > "$OUT/np.json" 2> "$OUT/np.err"
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d "$OUT/pl" -o pl --output-format csv -- $B > "$OUT/pl.json" 2> "$OUT/pl.err"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch --output-format csv -- $B --no-pipeline --steps 5 --warmup 1 > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write --output-format csv -- $B --no-pipeline --steps 5 --warmup 1 > "$OUT/write.json" 2> "$OUT/write.err"
# the rank kernel alone: memory-side read requests of the L2 by size (is a random 64-byte block read a 64-byte or a 128-byte request?)
R="python3 $ROOT/scripts/rank_only.py"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d "$OUT/rdreq" -o rdreq --output-format csv -- $R > "$OUT/rdreq.log" 2> "$OUT/rdreq.err" || echo "rdreq pass failed" >> "$OUT/rdreq.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d "$OUT/rdreq2" -o rdreq2 --output-format csv -- $R > "$OUT/rdreq2.log" 2> "$OUT/rdreq2.err" || echo "rdreq2 pass failed" >> "$OUT/rdreq2.err"
# the same split for the streaming scan and the context kernels (one short bench run)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d "$OUT/rdreq_bench" -o rdreq_bench --output-format csv -- $B --no-pipeline --steps 3 --warmup 1 --rank-queries 1048576 > "$OUT/rdreq_bench.json" 2> "$OUT/rdreq_bench.err" || echo "rdreq_bench pass failed" >> "$OUT/rdreq_bench.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" -o tcc --output-format csv -- $R > "$OUT/tcc.log" 2> "$OUT/tcc.err" || echo "tcc pass failed" >> "$OUT/tcc.err"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/rfetch" -o rfetch --output-format csv -- $R > "$OUT/rfetch.log" 2> "$OUT/rfetch.err" || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/rwrite" -o rwrite --output-format csv -- $R > "$OUT/rwrite.log" 2> "$OUT/rwrite.err" || true
# gather ceiling of the chip for independent random 64-byte block reads (scripts/rank_probe.hip), 4 GB table
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/rank_probe "$ROOT/scripts/rank_probe.hip" && timeout -k 10 200 /tmp/rank_probe 4096 > "$OUT/rank_probe.txt" 2>&1 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe "$ROOT/scripts/stream_probe.hip" && timeout -k 10 100 /tmp/stream_probe > "$OUT/stream_probe.txt" 2>&1 || true
# the per-dispatch tables are far too large to travel back (gpurun merges at most 64 MiB): summarise here, keep the small files
mkdir -p "$OUT/summary"
python3 "$ROOT/scripts/pmc_summary.py" "$TAG" "$OUT/summary" > "$OUT/summary/pmc_summary.log" 2>&1 || true
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*.csv' | head -30
du -sh "$OUT"
