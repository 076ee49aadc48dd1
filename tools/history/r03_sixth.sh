#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tests/golden/device_cases.py > gpurun_out/r03_device_cg_iterations.json 2> gpurun_out/r03_device_cases.err; echo "device_cases rc=$?"
cp gpurun_out/r03_device_cg_iterations.json tests/golden/device_cg_iterations.json
timeout -k 10 480 python -m pytest tests -m gpu -q --maxfail=20 > gpurun_out/r03_gpu_suite6.log 2>&1; echo "suite rc=$?"; tail -12 gpurun_out/r03_gpu_suite6.log
timeout -k 10 360 python bench.py > gpurun_out/r03_bench6.json 2> gpurun_out/r03_bench6.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench6.json'))
print('value', d['value'], 'single', d['single_walker']['sweeps_per_s'], 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
print('procs', [(p.get('procs'), round(p.get('sweeps_per_s',0),1), p.get('error')) for p in d['procs_per_gpu_scan']['points']])
print('threads', [(p['threads'], round(p['sweeps_per_s'],1)) for p in d['threads_per_gpu_scan']['points']])
r=d['roofline']; print('roofline frac', r['frac'], r['frac_single_pass'], r['frac_traffic'], r['avg_launch_us'])
print('scan', [(x['batch'], round(x['us'],1), round(x['frac'],2)) for x in r['batch_scan']])
print('cg_iteration_traffic', d.get('cg_iteration_traffic',{}).get('frac'))
PY
