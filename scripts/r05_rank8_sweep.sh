#!/bin/bash
# Round 5: a rank's share rehearsed on one GPU under the forms of the job's stream layout (EB_X_SPLIT: walk beside the extension) and
# of the scan's share of the CUs beside the context kernels (EB_SCAN_OVERLAP).  One line per form.  FORMS="a;b;c" with a = "label|ENV=.. ENV=.."
out=${1:-gpurun_out/r05_rank8_sweep.txt}
: > $out
run() {
  label=$1; shift
  env EB_BENCH_FORCE_DIST=1 EB_BENCH_TRACE=1 "$@" timeout -k 10 300 python bench.py --rehearse-rank-of ${RANK_OF:-8} --steps 200 --warmup 20 --cpu-sample-reads 0 --rank-queries 1048576 > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$label: FAILED" >> $out; tail -3 /tmp/sweep.err >> $out; return; }
  python - "$label" >> $out <<'PY'
import json, sys
d = json.load(open('/tmp/sweep.json'))
p = d['phase4_device_ms']
tr = [l for l in open('/tmp/sweep.err') if 'host trace, us' in l]
print(f"{sys.argv[1]:34s} ms/step {d['ms_per_step']:.4f} (median window {d['step_spread']['ms_per_step_median']:.4f}) | timed region: scan {p['scan']:.3f} extend {p['extend']:.3f} right {p['right']:.3f} span {p['context_span']:.3f} | alone: scan {p['alone']['scan_ms']:.3f} extend {p['alone']['extend_ms']:.3f} right {p['alone']['right_ms']:.3f} | {tr[-1].split('us: ')[1].strip() if tr else ''}")
PY
}
IFS=';' read -ra forms <<< "${FORMS:-default|;split|EB_X_SPLIT=1;overlap=4|EB_SCAN_OVERLAP=4;split overlap=4|EB_X_SPLIT=1 EB_SCAN_OVERLAP=4;split overlap=6|EB_X_SPLIT=1 EB_SCAN_OVERLAP=6;split overlap=3|EB_X_SPLIT=1 EB_SCAN_OVERLAP=3}"
for f in "${forms[@]}"; do
  label=${f%%|*}; envs=${f#*|}
  run "$label" $envs
done
cat $out
