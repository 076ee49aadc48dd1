#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_gpu_suite.log
WORKLOADS="holstein_honeycomb_L4_Ltau40 holstein_honeycomb_L8_Ltau80 bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100" bash tools/other_workloads.sh 2>&1 | grep -v "^$" | grep -v rc=
