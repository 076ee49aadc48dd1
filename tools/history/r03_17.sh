#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q --deselect tests/test_golden.py::test_device_iteration_counts_are_pinned > gpurun_out/r03_t17.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r03_t17.log
for nw in 1 16; do SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1; done
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b17.json 2>gpurun_out/r03_b17.err; python -c "import json; d=json.load(open('gpurun_out/r03_b17.json')); print('bench', round(d['value'],1))"; done
