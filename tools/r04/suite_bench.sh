#!/bin/bash
# full GPU suite + the default bench (+ the same with the generic force kernel) at HEAD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
for v in 1 0 1 0; do SMOQY_DMDX_FAST=$v timeout -k 10 300 python bench.py --timed-only --steps 10 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dmdx fast = $v:', round(d['value'],1), 'sweeps/s')"; done | tee gpurun_out/r04_dmdx_bench_ab.txt
