#!/bin/bash
# round 4, first GPU call: new sweep-parity tests, the whole suite, the default bench with the trajectory-following CPU baseline, one profiled run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_sweep_parity.py tests/test_bench_cpu_baseline.py -m gpu -x -q > gpurun_out/r04_new_tests.log 2>&1; echo new rc=$?; tail -5 gpurun_out/r04_new_tests.log
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
t0=$SECONDS; timeout -k 10 600 python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_output.json').read().strip().splitlines()[-1])
print(d['value'], d['avg_cg_iters'], d['cg_iterations_per_s'], d['cpu_baseline']['value'], d['cpu_baseline']['avg_cg_iters'], d['cpu_baseline']['cg_iterations_per_s'], d['single_walker']['sweeps_per_s'])
PY
bash tools/profile_bench.sh r04 && echo prof ok
