#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/stream_scan.py 1,2,4,8,32 0,2,4,8 2>&1 | tail -6
for R in 0 2 4 8 16; do
SMOQY_FDM_STREAM=$R timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b5_R$R.json 2>gpurun_out/r03_b5_R$R.err; python -c "import json; d=json.load(open('gpurun_out/r03_b5_R$R.json')); print('R=$R', round(d['value'],1), round(d['roofline']['avg_launch_us'],2))"
done
