#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_stream_mtm.py tests/test_gpu_bench_shape.py -m gpu -q -x > gpurun_out/r03_t50.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r03_t50.log
for os in 1 0; do echo "ownstream=$os"; SMOQY_FDM_OWNSTREAM=$os timeout -k 10 200 python tools/stream_scan.py 16,32,64,128 -1,2,4,8,16,32 2>&1 | tail -4; done
