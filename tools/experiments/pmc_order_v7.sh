#!/bin/bash
# FETCH_SIZE / TCC hit of the three work orders of tools/experiments/order_pair_v7.py (dispatch order = runs, pair per wg, XCD ranges)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_order; mkdir -p $OUT; export TMPDIR=/tmp
run() { local d=$OUT/$1; shift; rm -rf $d; (cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d $d -- python3 $ROOT/tools/experiments/order_pair_v7.py pmc) > $OUT/log_$(basename $d).txt 2>&1; echo "pass $(basename $d) done"; }
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_order/*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "pair_topk_kernel_v7" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    byd = {}
    for r in rows:
        byd.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for i, (d, c) in enumerate(sorted(byd.items())):
        print(f, "dispatch", d, "mode", i % 6, {k: (2 * v * 1024 / 1e9 if k == "FETCH_SIZE" else v) for k, v in c.items()})
PY
