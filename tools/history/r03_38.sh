#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_gpu_suite.log
timeout -k 10 500 python tools/team_procs_scan.py 32,64 4 2>&1 | tail -2 | cut -c1-160
timeout -k 10 500 python tools/team_scan.py 64 4 4x32 2>&1 | tail -2 | cut -c1-330
