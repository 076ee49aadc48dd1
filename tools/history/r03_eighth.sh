#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_bench.sh r03mid; head -14 gpurun_out/r03mid_bench_kernel_stats.txt
bash tools/config_scan.sh "128 8 1 3" "128 4 1 3" "128 4 1 2" "192 6 1 3" "128 8 1 4" "256 8 1 3" 2>&1 | tail -16
