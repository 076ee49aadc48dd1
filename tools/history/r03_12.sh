#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_stream_mtm.py -m gpu -q -x > gpurun_out/r03_stream_test.log 2>&1; echo "stream test rc=$?"; tail -3 gpurun_out/r03_stream_test.log
timeout -k 10 300 python tools/stream_scan.py 16,32,64,128 0,2,4,8,16,32 2>&1 | tail -5
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b12_$i.json 2>gpurun_out/r03_b12_$i.err; python -c "import json; d=json.load(open('gpurun_out/r03_b12_$i.json')); print('bench', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"; done
