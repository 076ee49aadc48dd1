#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_gpu_suite.log
for rep in 1 2 3; do for os in 1 0; do SMOQY_FDM_OWNSTREAM=$os timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b51.json 2>gpurun_out/r03_b51.err; python -c "import json; d=json.load(open('gpurun_out/r03_b51.json')); print('bench ownstream=$os', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"; done; done
for os in 1 0; do for nw in 16 64; do echo "ownstream=$os nw=$nw: $(SMOQY_FDM_OWNSTREAM=$os SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done
