#!/bin/bash
# rocprofv3 --pmc passes over the two pair kernels (counters only; the interpreter directly after `--`)
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_pair; mkdir -p $OUT; export TMPDIR=/tmp
run() { local d=$OUT/$1; shift; rm -rf $d; (cd /tmp && rocprofv3 --pmc "$@" --output-format csv -d $d -- python3 $ROOT/tools/experiments/run_pair_once.py) > $OUT/log_$(basename $d).txt 2>&1; echo "pass $(basename $d) done"; }
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
run wait SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES
run busy SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU
run valu SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("OUT", "gpurun_out/pmc_pair")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pair_topk" not in k: continue
        agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):.4g}  (n={len(v)})")
PY
