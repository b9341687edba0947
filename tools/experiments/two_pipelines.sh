#!/bin/bash
# Is there capacity left on the chip beside ONE pipeline of clips?  Two bench processes on the same GPU at once against one alone.
F="--steps 400 --warmup 20 --repeats 0 --no-cpu-baseline --no-corr-volume --no-f16x3-line --no-clips-line"
one() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1], round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms/step")
PY
}
timeout -k 10 300 python3 bench.py $F > gpurun_out/solo.json 2> gpurun_out/solo.err && one gpurun_out/solo.json
timeout -k 10 300 python3 bench.py $F > gpurun_out/duo_a.json 2> gpurun_out/duo_a.err &
PA=$!
timeout -k 10 300 python3 bench.py $F > gpurun_out/duo_b.json 2> gpurun_out/duo_b.err
wait $PA
one gpurun_out/duo_a.json; one gpurun_out/duo_b.json
