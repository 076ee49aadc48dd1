#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_c_abi.py -m gpu -q -x > gpurun_out/r03_t29.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r03_t29.log
timeout -k 10 300 python examples/walker_team_ranks.py holstein_honeycomb_L16_Ltau128 32 4 2>&1 | tail -4
