#!/bin/bash
# the in-tree build at the round's last commit: GPU suite, smoke(), default bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo smoke rc=$?; tail -1 gpurun_out/r04_smoke.log
t0=$SECONDS; timeout -k 10 600 python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_output.json').read().strip().splitlines()[-1])
print('sweeps/s', d['value'], 'iters', d['avg_cg_iters'], 'cpu', d['cpu_baseline']['value'], 'single', d['single_walker']['sweeps_per_s'], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1))
PY
