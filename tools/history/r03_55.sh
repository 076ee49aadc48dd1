#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r03_1walker 1 && SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r03_16walkers 16
tail -16 gpurun_out/gap_r03_16walkers.txt; tail -14 gpurun_out/gap_r03_1walker.txt
bash tools/config_scan.sh "128 8 1 4" "128 8 1 3" "128 8 1 5" "128 4 1 4" "128 4 1 0" "96 6 1 4" "192 8 1 4" "256 8 1 4" > gpurun_out/r03_config_scan.txt 2>&1; cat gpurun_out/r03_config_scan.txt
