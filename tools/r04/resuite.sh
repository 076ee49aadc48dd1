#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -6 gpurun_out/r04_gpu_suite.log
timeout -k 10 600 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.new.json 2> gpurun_out/device_cases.err; echo pins rc=$?
python - <<'PY'
import json
old=json.load(open('tests/golden/device_cg_iterations.json')); new=json.load(open('gpurun_out/device_cg_iterations.new.json'))
for k in sorted(new['device']):
    a,b=old['device'].get(k),new['device'][k]
    if a!=b:
        d=[y-x for x,y in zip(a,b)] if a else None
        print('moved', k, 'max |delta|', max(abs(v) for v in d) if d else None)
PY
