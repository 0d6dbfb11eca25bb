#!/bin/bash
# Round 5: GPU_MAX_HW_QUEUES (hardware queues the HIP runtime opens for the process; rounds 2-4 ran with 16) against the runtime's
# default of 4, on a rank's eighth (rehearsal) and on the whole range (N = 1).
out=${1:-gpurun_out/r05_hwq_ab.txt}
: > $out
for q in 4 16 6 4 16; do
  GPU_MAX_HW_QUEUES=$q EB_BENCH_FORCE_DIST=1 EB_BENCH_TRACE=1 timeout -k 10 300 python bench.py --rehearse-rank-of 8 --steps 200 --warmup 20 --cpu-sample-reads 0 --rank-queries 1048576 > /tmp/h.json 2> /tmp/h.err || { echo "queues=$q rehearsal: FAILED" >> $out; continue; }
  python - "$q" "rank's eighth" >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/h.json')); p = d['phase4_device_ms']
print(f"GPU_MAX_HW_QUEUES={sys.argv[1]:>2s} {sys.argv[2]:14s} ms/step {d['ms_per_step']:.4f} (median window {d['step_spread']['ms_per_step_median']:.4f}) | timed region: scan {p['scan']:.3f} extend {p['extend']:.3f} right {p['right']:.3f} span {p['context_span']:.3f} | alone: scan {p['alone']['scan_ms']:.3f} extend {p['alone']['extend_ms']:.3f} right {p['alone']['right_ms']:.3f}")
PY
done
for q in 4 16 4 16; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-sample-reads 0 --rank-queries 1048576 > /tmp/h.json 2> /tmp/h.err || { echo "queues=$q whole range: FAILED" >> $out; continue; }
  python - "$q" "whole range" >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/h.json')); p = d['phase4_device_ms']
print(f"GPU_MAX_HW_QUEUES={sys.argv[1]:>2s} {sys.argv[2]:14s} ms/step {d['ms_per_step']:.4f} (median window {d['step_spread']['ms_per_step_median']:.4f}) | timed region: scan {p['scan']:.3f} extend {p['extend']:.3f} right {p['right']:.3f} span {p['context_span']:.3f} | alone: scan {p['alone']['scan_ms']:.3f} extend {p['alone']['extend_ms']:.3f} right {p['alone']['right_ms']:.3f} | without the FL table {d['fl_table']['ms_per_step_without_it']} | traversal {d['phases23_gpu']['seconds']}")
PY
done
cat $out
