#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r03_gpu_suite2.log 2>&1; echo "suite rc=$?"; tail -15 gpurun_out/r03_gpu_suite2.log
SMOQY_EFA=1 bash tools/gap_probe.sh r03_1walker_efa 1; cat gpurun_out/gap_r03_1walker_efa.txt
WORKLOADS="ossh_square_L12_Ltau100 bssh_chain_L256_Ltau200" STEPS=2 WARMUP=1 bash tools/other_workloads.sh
