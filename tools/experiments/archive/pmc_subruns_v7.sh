#!/bin/bash
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sub; mkdir -p $OUT; export TMPDIR=/tmp
d=$OUT/fetch; rm -rf $d
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 $ROOT/tools/experiments/subruns_pair_v7.py pmc) > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
for f in sorted(glob.glob("gpurun_out/pmc_sub/fetch/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "pair_topk_kernel_v7" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for i, r in enumerate(rows[-8:]):
        print("dispatch", r["Dispatch_Id"], "mode", (len(rows) - 8 + i) % 4, "L2 fills GB", round(2 * float(r["Counter_Value"]) * 1024 / 1e9, 3))
PY
