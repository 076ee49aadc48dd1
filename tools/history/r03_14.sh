#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "--tfft-form in-place" "--tfft-form two-image" "--tfft-form in-place --cg-split 0" "--tfft-form in-place --solve-concurrency 6" "--tfft-form in-place --streams 16 --solve-concurrency 4" ; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 --no-mtm-sampling $cfg > gpurun_out/r03_b14.json 2>gpurun_out/r03_b14.err; python -c "import json; d=json.load(open('gpurun_out/r03_b14.json')); print('$cfg', round(d['value'],1))"
done; done
