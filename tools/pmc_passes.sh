#!/bin/bash
# The four rocprofv3 --pmc passes of tools/run_kernels_once.py (counters only: never combined with a trace; the interpreter directly
# after `--`) and the report:   bash tools/pmc_passes.sh r05   ->   gpurun_out/r05_pmc.json
set -e
#     bash tools/pmc_passes.sh r06_cfg3 tools/run_cfg3_once.py   ->   the same for the configs[2] kernels
TAG=${1:-r05}; SCRIPT=${2:-tools/run_kernels_once.py}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { local d=$OUT/$1; shift; rm -rf $d; (cd /tmp && rocprofv3 --pmc "$@" -d $d -- python3 $ROOT/$SCRIPT) > $OUT/log_$(basename $d).txt 2>&1; echo "pass $(basename $d) done"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
python3 tools/pmc_report.py $OUT/fetch $OUT/write $OUT/tcc $OUT/mfma > $ROOT/gpurun_out/${TAG}_pmc.json
rm -rf $OUT/fetch $OUT/write $OUT/tcc $OUT/mfma      # (raw counter databases: tens of MB each; gpurun_out is copied back only below 64 MiB)
echo "wrote gpurun_out/${TAG}_pmc.json"
