#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python tools/team_scan.py 16,32 4 2x32,4x32,8x16 2>&1 | tee gpurun_out/r03_team_scan_devhmc.txt | tail -8
timeout -k 10 500 python tools/team_procs_scan.py 16,32,64,2x32,4x32 4 2>&1 | tee gpurun_out/r03_team_procs_devhmc.txt | tail -8
