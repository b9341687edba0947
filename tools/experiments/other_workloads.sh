#!/bin/bash
# the bench's other workloads (BASELINE configs[0], [3], [4] shapes) + small-clip variants; one line each -> stdout
for args in "--workload cfg1_256_2f" "--workload cfg1_256_2f --encoder-lanes 1" "--workload cfg1_256_2f --encoder-graph" "--workload cfg1_256_2f --encoder-lanes 1 --encoder-graph" "--workload cfg4_davis_64f" "--workload cfg5_720p_24f"; do
  python bench.py $args --steps 30 --repeats 1 --no-cpu-baseline --no-corr-volume --no-f16x3-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', '->', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms/step', {k: round(v,3) for k,v in d.get('sharding_ms_per_step',{}).items() if v>0.01}, 'pair', d['kernels'].get('pair_topk',{}).get('kernel'), round(d['kernels'].get('pair_topk',{}).get('ms_per_launch',0),3))"
done
