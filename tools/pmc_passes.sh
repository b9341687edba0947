#!/bin/bash
# Four separate rocprofv3 --pmc passes of tools/run_kernels_once.py (counters are never combined with traces; the interpreter
# sits directly after `--`), then tools/pmc_report.py -> profiles/<name>.  Run from the repository root on the GPU box:
#     bash tools/pmc_passes.sh gpurun_out/pmc r04_pmc.json
set -e
OUT=${1:-gpurun_out/pmc}
NAME=${2:-r04_pmc.json}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p "$OUT"
run() {  # dir, counters...
  local d="$ROOT/$OUT/$1"; shift
  rm -rf "$d"
  (cd /tmp && rocprofv3 --pmc "$@" -d "$d" -- python3 "$ROOT/tools/run_kernels_once.py") > "$ROOT/$OUT/pass_$(basename $d).log" 2>&1
  echo "pass $(basename $d) done"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
python3 tools/pmc_report.py "$OUT/fetch" "$OUT/write" "$OUT/tcc" "$OUT/mfma" > "$OUT/$NAME"
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/tcc" "$OUT/mfma"     # raw databases: tens of MB each
echo "wrote $OUT/$NAME"
