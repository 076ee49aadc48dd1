#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_team.py -m gpu -q -x > gpurun_out/r03_t19.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_t19.log
timeout -k 10 500 python tools/team_scan.py 8,16,32 4 2>&1 | tee gpurun_out/r03_team_scan.txt | tail -5
