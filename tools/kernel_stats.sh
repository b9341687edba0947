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
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/raw" -- python3 "$ROOT/bench.py" --steps 40 --repeats 1 --no-cpu-baseline --no-f16x3-line --no-clips-line) > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
f=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/$NAME"
# ... and per (symbol, grid): one template instance serves launches of different sizes (conv256p_kernel<256, 3, false, false, ...> is the
# 128 -> 256 AND the 256 -> 256 convolution), which the per-symbol averages above mix (round-5 review, "what's weak" 6)
t=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 - "$t" "$OUT/${NAME%.csv}_by_grid.csv" <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
col = lambda r, *names: next((r[n] for n in names if n in r), "")
acc = collections.defaultdict(list)
for r in rows:
    name = col(r, "Kernel_Name", "Name")
    grid = "x".join(col(r, f"Grid_Size_{a}", f"Grid_Size{a}") or "?" for a in "XYZ")
    wg = "x".join(col(r, f"Workgroup_Size_{a}", f"Workgroup_Size{a}") or "?" for a in "XYZ")
    acc[(name, grid, wg)].append(int(col(r, "End_Timestamp")) - int(col(r, "Start_Timestamp")))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, grid, wg, len(v), sum(v), sum(v) / len(v), min(v), max(v)])
PY
rm -rf "$OUT/raw"
echo "wrote $OUT/$NAME and $OUT/${NAME%.csv}_by_grid.csv"
