#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -m gpu -q -x > gpurun_out/r03_t47.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03_t47.log
bash tools/solo_profile.sh r03_pfx 16 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_pfx.txt | cut -c1-150
bash tools/solo_profile.sh r03_pfx_w1 1 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_pfx_w1.txt | cut -c1-150
