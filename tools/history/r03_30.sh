#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q -x -k "timing or iterate" > gpurun_out/r03_t30.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_t30.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-proc-scan --steps 4 --warmup 2 > gpurun_out/r03_b30.json 2> gpurun_out/r03_b30.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_b30.json'))
print(d['value'])
print(json.dumps(d['iteration_kernels'], indent=1)[:3000])
PY
