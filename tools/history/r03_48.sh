#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo rc=$?; tail -4 gpurun_out/r03_gpu_suite.log
bash tools/solo_profile.sh r03_hc16 16 > /dev/null 2>&1; head -9 gpurun_out/solo_r03_hc16.txt | cut -c1-150
bash tools/solo_profile.sh r03_hc16_w1 1 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_hc16_w1.txt | cut -c1-150
SMOQY_CHEB_WL0=1 bash tools/solo_profile.sh r03_hc16_w1_bperm 1 > /dev/null 2>&1; head -5 gpurun_out/solo_r03_hc16_w1_bperm.txt | grep cheb | cut -c1-150
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b48.json 2>gpurun_out/r03_b48.err; python -c "import json; d=json.load(open('gpurun_out/r03_b48.json')); print('bench', round(d['value'],1))"; done
for nw in 1 16; do echo "nw=$nw: $(SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done
