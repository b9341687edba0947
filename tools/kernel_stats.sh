#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command -> <out>/<name> (the kernel_stats CSV) + the bench's JSON line.
#     bash tools/kernel_stats.sh gpurun_out/stats r04_bench_kernel_stats.csv
set -e
OUT=${1:-gpurun_out/stats}
NAME=${2:-r04_bench_kernel_stats.csv}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"
rm -rf "$OUT/raw"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/raw" -- python3 "$ROOT/bench.py" --steps 40 --repeats 1 --no-cpu-baseline) > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
f=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/$NAME"
rm -rf "$OUT/raw"
echo "wrote $OUT/$NAME"
