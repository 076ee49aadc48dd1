#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_efa.py tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -q > gpurun_out/r03_t15.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r03_t15.log
for nw in 1 16; do SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1; done
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b15.json 2>gpurun_out/r03_b15.err; python -c "import json; d=json.load(open('gpurun_out/r03_b15.json')); print('bench', round(d['value'],1))"; done
