#!/bin/bash
# round 3, first GPU pass: regenerate the device iteration pins (new alpha = 1 shapes), full GPU suite, default bench with the new scans
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tests/golden/device_cases.py > gpurun_out/r03_device_cg_iterations.json 2> gpurun_out/r03_device_cases.err; echo "device_cases rc=$?"
cp gpurun_out/r03_device_cg_iterations.json tests/golden/device_cg_iterations.json
timeout -k 10 480 python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_suite1.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r03_gpu_suite1.log
timeout -k 10 360 python bench.py > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench1.json'))
print('value', d['value'], 'single', d['single_walker']['sweeps_per_s'], 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
print('procs', [(p.get('procs'), round(p.get('sweeps_per_s',0),1), p.get('error')) for p in d['procs_per_gpu_scan']['points']])
print('threads', d['threads_per_gpu_scan'])
print('roofline frac', d['roofline']['frac'], d['roofline']['frac_single_pass'], d['roofline']['frac_traffic'], d['roofline']['avg_launch_us'])
print('cg_iteration_traffic', d.get('cg_iteration_traffic'))
PY
