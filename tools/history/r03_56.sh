#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_team.py -m gpu -q -x 2>&1 | tail -5
cat /proc/loadavg
for pf in 1 0; do echo "prefetch=$pf"; SMOQY_PREFETCH=$pf timeout -k 10 500 python tools/team_scan.py 16,32,64 4 2x32,4x32 2>&1 | cut -c1-330; done
for pf in 1 0; do echo "procs prefetch=$pf"; SMOQY_PREFETCH=$pf timeout -k 10 500 python tools/team_procs_scan.py 16,32,64,2x32 4 2>&1 | cut -c1-200; done
