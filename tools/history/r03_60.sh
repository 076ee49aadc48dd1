#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cat /proc/loadavg
for rep in 1 2; do for os in 1 0; do
  echo "ownstream=$os nw=32: $(SMOQY_FDM_OWNSTREAM=$os timeout -k 10 120 python tools/history/one_stream_iters.py 32 2>&1 | tail -1)"
  echo "ownstream=$os nw=64: $(SMOQY_FDM_OWNSTREAM=$os timeout -k 10 120 python tools/history/one_stream_iters.py 64 2>&1 | tail -1)"
  echo "ownstream=$os 4x32 native: $(SMOQY_FDM_OWNSTREAM=$os timeout -k 10 300 python tools/team_scan.py 64 4 4x32 2>&1 | grep -o 'native_threads_sweeps_per_s": [0-9.]*' | tr '\n' ' ')"
  SMOQY_FDM_OWNSTREAM=$os timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --walkers-per-gpu 256 --streams 8 > gpurun_out/r03_b60.json 2>gpurun_out/r03_b60.err; python -c "import json; d=json.load(open('gpurun_out/r03_b60.json')); print('bench 256/8 ownstream=$os', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
done; done
