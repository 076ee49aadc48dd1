#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python tools/team_scan.py 64 4 2x32,4x32,8x16,4x16,2x64 2>&1 | tee gpurun_out/r03_team_multi.txt | tail -8
