#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_suite2.log 2>&1; tail -3 gpurun_out/r02_gpu_suite2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 > gpurun_out/poll1.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/poll1.json')); print('poll', round(d['value'],1), [round(x['sweeps_per_s'],1) for x in d['one_stream']])
PY
