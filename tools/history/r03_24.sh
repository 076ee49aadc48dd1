#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sb in 8 16 4 8 16; do SMOQY_TFFT_SB=$sb timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b24.json 2>gpurun_out/r03_b24.err; python -c "import json; d=json.load(open('gpurun_out/r03_b24.json')); print('bench SB=$sb', round(d['value'],1))"; done
for sb in 8 16; do for nw in 1 16; do echo "SB=$sb nw=$nw"; SMOQY_TFFT_SB=$sb SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1; done; done
