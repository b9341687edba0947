#!/bin/bash
# rebuilds pair_v7.inc with one bisection toggle at a time and runs the parity check
for t in "" f0early qscres nopeek headall "f0early,qscres,nopeek,headall"; do
  V7_GEN_DBG="$t" python tools/gen_pair_v7.py > /dev/null && python -m fgvc_amd.build > /dev/null 2>&1
  echo "=== toggles: [$t]"
  timeout -k 10 120 python tools/experiments/dbg_pair_f16f6.py 2>&1 | grep -E "debug 0 runs True|debug 16384 runs True|timed" | head -3
done
python tools/gen_pair_v7.py > /dev/null
