#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_team.py tests/test_native_randn.py -m "gpu or not gpu" -q -x 2>&1 | tail -3
cat /proc/loadavg
timeout -k 10 500 python tools/team_scan.py 16,32,64 4 2x32,4x32,8x16 2>&1 | cut -c1-330
cat /proc/loadavg
