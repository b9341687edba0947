#!/bin/bash
# A/B/A/B of two builds of the library on ONE box through bench.py (boxes of the pool differ by up to 8 %):
#     bash tools/ab_bench.sh fgvc_amd/lib/libfgvc_hip_prev.so [extra bench flags]   ->  gpurun_out/ab_bench.log
ALT=$1; shift
OUT=gpurun_out/ab_bench.log; mkdir -p gpurun_out; : > $OUT
for rep in 1 2; do
  for lib in "" "$ALT"; do
    FGVC_HIP_LIB=$lib timeout -k 10 300 python3 bench.py --steps 60 --repeats 1 --no-cpu-baseline --no-corr-volume --no-f16x3-line --no-clips-line "$@" > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "bench failed (lib=$lib)" >> $OUT; tail -3 gpurun_out/ab_one.err >> $OUT; continue; }
    python3 - "$lib" >> $OUT <<'PY'
import json, sys
d = json.loads([l for l in open("gpurun_out/ab_one.json") if l.startswith("{")][-1])
r, k = d["roofline"], d.get("kernels", {})
print(f"{sys.argv[1] or 'default':40s} {d['value']:8.1f} frames/s  {d['ms_per_step']:.3f} ms/step  conv256 {r.get('ms_per_launch'):.4f} ms/launch frac {r.get('frac'):.4f}  "
      f"encode {d.get('sharding_ms_per_step', {}).get('encode', 0):.3f} ms  pair {k.get('pair_topk', {}).get('ms_per_launch', 0):.3f} ms")
PY
  done
done
cat $OUT
