#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for cfg in "128 8 4" "128 8 3" "128 8 5" "128 8 6" "256 8 4" "192 12 5"; do set -- $cfg; timeout -k 10 300 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 --walkers-per-gpu $1 --streams $2 --solve-concurrency $3 > gpurun_out/r03_b39.json 2>gpurun_out/r03_b39.err; python -c "import json; d=json.load(open('gpurun_out/r03_b39.json')); print('walkers $1 streams $2 gate $3:', round(d['value'],1))"; done; done
