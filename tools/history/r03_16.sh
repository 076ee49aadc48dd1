#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_stream_mtm.py -m gpu -q -x > gpurun_out/r03_stream_test.log 2>&1; echo "stream test rc=$?"; tail -3 gpurun_out/r03_stream_test.log
timeout -k 10 200 python tools/stream_scan.py 16,64 0,2,4,8,16 ossh_square_L12_Ltau100 2>&1 | tail -3
timeout -k 10 200 python tools/stream_scan.py 16,64 0,2,4,8,16 bssh_chain_L256_Ltau200 2>&1 | tail -3
timeout -k 10 200 python tools/stream_scan.py 16,64 0,2,4,8,16 holstein_honeycomb_L8_Ltau80 2>&1 | tail -3
