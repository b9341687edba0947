#!/bin/bash
# A/B/A/B of bench flag sets on one box:  bash tools/ab_flags.sh "<flags A>" "<flags B>" ...  -> gpurun_out/ab_flags.log
OUT=gpurun_out/ab_flags.log; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for fl in "$@"; do
    timeout -k 10 300 python3 bench.py --steps 60 --repeats 1 --no-cpu-baseline --no-corr-volume --no-f16x3-line --no-clips-line $fl > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "bench failed (flags=$fl)" >> $OUT; tail -3 gpurun_out/ab_one.err >> $OUT; continue; }
    python3 - "$fl" >> $OUT <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_one.json") if l.startswith("{")][-1])
r = d["roofline"]
print(f"{sys.argv[1] or '(default)':40s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.3f} ms/step  conv256 {r.get('ms_per_launch'):.4f} ms/launch  blocks {d.get('step_ms_min')}-{d.get('step_ms_max')}")
PY
  done
done
cat $OUT
