#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SMOQY_XCD_MAP=1 timeout -k 10 300 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r03_t41.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_t41.log
for rep in 1 2; do for xm in 0 1; do for nw in 8 16 32; do echo "xcd_map=$xm nw=$nw: $(SMOQY_XCD_MAP=$xm SMOQY_EFA=1 SMOQY_SPLIT=1 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done; done
SMOQY_XCD_MAP=1 bash tools/solo_profile.sh r03_xcd1 16 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_xcd1.txt | cut -c1-150
for rep in 1 2 3; do for xm in 0 1; do SMOQY_XCD_MAP=$xm timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b41.json 2>gpurun_out/r03_b41.err; python -c "import json; d=json.load(open('gpurun_out/r03_b41.json')); print('bench xcd_map=$xm', round(d['value'],1))"; done; done
