#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/config_scan.sh "128 8 1 3" "128 8 1 4" "128 8 1 5" "256 8 1 4" "256 8 1 3" "192 8 1 4" 2>&1 | tail -16
