#!/bin/bash
# Round-6 evidence in one GPU session: the whole GPU test suite, the default bench line, rocprofv3 kernel stats of the same command, the
# other BASELINE workloads, and the configs[2] kernel trace + counter passes.  Everything lands in gpurun_out/ (raw traces are deleted:
# the directory comes back only below 64 MiB).      bash tools/evidence_r06.sh
mkdir -p gpurun_out
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > gpurun_out/r06_gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a gpurun_out/r06_gpu_tests.log
tail -3 gpurun_out/r06_gpu_tests.log
timeout -k 10 600 python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"
bash tools/kernel_stats.sh gpurun_out/stats r06_bench_kernel_stats.csv > gpurun_out/kernel_stats.log 2>&1; echo "kernel stats rc=$?"
bash tools/experiments/other_workloads.sh > gpurun_out/r06_other_workloads.log 2>&1; echo "other workloads rc=$?"
bash tools/profile_cfg3.sh r06 > gpurun_out/profile_cfg3.log 2>&1; echo "cfg3 trace rc=$?"
bash tools/pmc_passes.sh r06_cfg3 tools/run_cfg3_once.py > gpurun_out/pmc_r06_cfg3.log 2>&1; echo "cfg3 pmc rc=$?"
du -sh gpurun_out
