#!/bin/bash
# configs[2] evidence: tools/bench_cfg3.py plain, then under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_cfg3_*
#     bash tools/profile_cfg3.sh r06
set -e
TAG=${1:-r06}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/bench_cfg3.py > $OUT/${TAG}_cfg3_bench.log 2>&1
rm -rf $OUT/cfg3_raw
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3_raw -- python3 $ROOT/tools/bench_cfg3.py) > $OUT/${TAG}_cfg3_bench_under_rocprof.log 2> $OUT/cfg3_rocprof.err
cp $(find $OUT/cfg3_raw -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg3_kernel_stats.csv
rm -rf $OUT/cfg3_raw
echo "wrote $OUT/${TAG}_cfg3_kernel_stats.csv"
