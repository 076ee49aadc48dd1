#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "128 8 4" "128 4 4" "128 4 2" "128 2 2" "256 8 4" "256 4 4" "256 4 2" "128 8 4"; do set -- $cfg; timeout -k 10 300 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 --walkers-per-gpu $1 --streams $2 --solve-concurrency $3 > gpurun_out/r03_b33.json 2>gpurun_out/r03_b33.err; python -c "import json; d=json.load(open('gpurun_out/r03_b33.json')); print('walkers $1 streams $2 gate $3:', round(d['value'],1))"; done
