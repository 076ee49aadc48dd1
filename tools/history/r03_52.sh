#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for os in 1 0; do for nw in 1 16 64; do echo "ownstream=$os nw=$nw: $(SMOQY_FDM_OWNSTREAM=$os timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; done; done; done
for rep in 1 2; do for os in 1 0; do SMOQY_FDM_OWNSTREAM=$os timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only > gpurun_out/r03_b52.json 2>gpurun_out/r03_b52.err; python -c "import json; d=json.load(open('gpurun_out/r03_b52.json')); print('bench ownstream=$os', round(d['value'],1), round(d['roofline']['avg_launch_us'],1), d['steps'])"; done; done
