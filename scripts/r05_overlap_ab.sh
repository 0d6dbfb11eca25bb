#!/bin/bash
# Round 5: the scan's share of each CU beside the previous pass's context kernels (EB_SCAN_OVERLAP workgroups per CU) on the BASELINE
# workload (error-free reads) and on reads with 1 % substitution errors, N = 1, whole range.
out=${1:-gpurun_out/r05_overlap_ab.txt}
: > $out
for extra in "" "--error-rate-ppm 10000"; do
for ov in 8 6 5 4; do
  EB_SCAN_OVERLAP=$ov timeout -k 10 300 python bench.py --steps 40 --warmup 5 --cpu-sample-reads 0 --rank-queries 1048576 $extra > /tmp/ov.json 2> /tmp/ov.err || { echo "overlap=$ov $extra: FAILED" >> $out; continue; }
  python - "$ov" "$extra" >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/ov.json'))
p = d['phase4_device_ms']
print(f"overlap={sys.argv[1]} {sys.argv[2] or 'error-free':24s} ms/step {d['ms_per_step']:.4f} (median window {d['step_spread']['ms_per_step_median']:.4f}) | timed region: scan {p['scan']:.3f} extend {p['extend']:.3f} right {p['right']:.3f} span {p['context_span']:.3f} | alone: scan {p['alone']['scan_ms']:.3f} extend {p['alone']['extend_ms']:.3f} right {p['alone']['right_ms']:.3f} | without the FL table {d['fl_table']['ms_per_step_without_it']}")
PY
done
done
cat $out
