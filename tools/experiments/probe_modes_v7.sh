#!/bin/bash
for cfg in "4:nofp6" "4:"; do
  ring=${cfg%%:*}; dbg=${cfg##*:}
  V7_GEN_RING=$ring V7_GEN_DBG=$dbg python tools/gen_pair_v7.py > /dev/null && python -m fgvc_amd.build > /dev/null 2>&1
  echo "=== ring $ring [$dbg]"
  timeout -k 10 120 python tools/experiments/probe_modes_v7.py 2>&1 | tail -4
done
python tools/gen_pair_v7.py > /dev/null
