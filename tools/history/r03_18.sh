#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --deselect tests/test_golden.py::test_device_iteration_counts_are_pinned > gpurun_out/r03_t18.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r03_t18.log
for xd in 1 0; do for nw in 1 16; do echo "xdefer=$xd nw=$nw"; SMOQY_CG_XDEFER=$xd SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1; done; done
for xd in 1 0 1 0; do SMOQY_CG_XDEFER=$xd timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b18.json 2>gpurun_out/r03_b18.err; python -c "import json; d=json.load(open('gpurun_out/r03_b18.json')); print('bench xdefer=$xd', round(d['value'],1))"; done
