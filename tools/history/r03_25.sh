#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3 4 5; do for sb in 8 16; do SMOQY_TFFT_SB=$sb timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b25.json 2>gpurun_out/r03_b25.err; python -c "import json; d=json.load(open('gpurun_out/r03_b25.json')); print('bench SB=$sb', round(d['value'],1))"; done; done
