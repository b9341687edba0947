# A/B runs of bench.py on ONE box (boxes differ by up to 8 %): each line is a variant against the default, interleaved three times.
set -e
for round in 1 2 3; do
for args in "" "--no-conv64-f16f8"; do
  python bench.py --steps 60 --repeats 2 --no-cpu-baseline --no-corr-volume --no-clips-line $args > gpurun_out/bv.json 2>/dev/null
  python - "$args" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/bv.json').read().strip().splitlines()[-1])
print(sys.argv[1] or "default", round(d["value"],1), "fps", round(d["ms_per_step"],3), "ms", {k:round(v,2) for k,v in d["sharding_ms_per_step"].items() if v>0.05}, flush=True)
PY
done
done
