#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2 3 4; do for xs in 0 1; do SMOQY_NT_FIELDS=$xs timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 8 --warmup 2 > gpurun_out/r03_b40.json 2>gpurun_out/r03_b40.err; python -c "import json; d=json.load(open('gpurun_out/r03_b40.json')); print('bench nt_fields=$xs', round(d['value'],1))"; done; done
