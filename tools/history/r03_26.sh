#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_team.py -m gpu -q -x > gpurun_out/r03_t26.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r03_t26.log
