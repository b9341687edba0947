#!/bin/bash
# ablation builds of fgvc_pair_topk_f16f6 (results WRONG where noted): how does the consumer's chain scale with the depth of its
# fragment ring?  "nofp6" drops the FP6 half (reads, conversions, MFMAs) so that a deeper ring fits the registers.
for cfg in "4:" "4:nofp6" "6:nofp6" "8:nofp6" "12:nofp6"; do
  ring=${cfg%%:*}; dbg=${cfg##*:}
  V7_GEN_RING=$ring V7_GEN_DBG=$dbg python tools/gen_pair_v7.py > /dev/null && python -m fgvc_amd.build > /dev/null 2>&1
  echo "=== ring $ring [$dbg]"
  timeout -k 10 120 python tools/experiments/time_only_f16f6.py 2>&1 | tail -1
done
python tools/gen_pair_v7.py > /dev/null
