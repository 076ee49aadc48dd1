#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_team.py -m gpu -q -x > gpurun_out/r03_t21.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r03_t21.log
timeout -k 10 600 python tools/team_procs_scan.py 8,16,32,64,2x32,4x32,4x16 4 2>&1 | tee gpurun_out/r03_team_procs.txt | tail -8
ls /dev/shm | head
