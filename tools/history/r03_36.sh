#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3 4; do for xs in 0 1; do SMOQY_X_STREAM=$xs timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b36.json 2>gpurun_out/r03_b36.err; python -c "import json; d=json.load(open('gpurun_out/r03_b36.json')); print('bench x_stream=$xs', round(d['value'],1))"; done; done
for xs in 0 1; do for nw in 1 16; do echo "x_stream=$xs nw=$nw: $(SMOQY_X_STREAM=$xs SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done
