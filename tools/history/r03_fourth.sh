#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_stream_mtm.py -m gpu -q -x > gpurun_out/r03_stream_test.log 2>&1; echo "stream test rc=$?"; tail -15 gpurun_out/r03_stream_test.log
timeout -k 10 300 python tools/stream_scan.py 2>&1 | tail -5
