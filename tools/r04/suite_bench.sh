#!/bin/bash
# full GPU suite (stops at the first failure), regenerated pins beside the old ones, default bench, three other workloads (short)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -12 gpurun_out/r04_gpu_suite.log
timeout -k 10 600 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.new.json 2> gpurun_out/device_cases.err; echo pins rc=$?
python - <<'PY'
import json
old=json.load(open('tests/golden/device_cg_iterations.json')); new=json.load(open('gpurun_out/device_cg_iterations.new.json'))
for k in sorted(new['device']):
    a,b=old['device'].get(k),new['device'][k]
    if a!=b:
        d=[y-x for x,y in zip(a,b)] if a else None
        print('moved', k, 'max |delta|', max(abs(v) for v in d) if d else None, 'oracle', new['oracle'].get(k))
PY
t0=$SECONDS; timeout -k 10 600 python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_output.json').read().strip().splitlines()[-1])
print('sweeps/s', d['value'], 'iters', d['avg_cg_iters'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['avg_cg_iters'], 'single', d['single_walker']['sweeps_per_s'], 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
print('roofline', {k: d['roofline'][k] for k in ('frac','frac_single_pass','frac_traffic','avg_launch_us')}, 'hbm', d['roofline']['hbm_resident_point']['us'])
PY
for wl in ossh_square_L12_Ltau100 bssh_chain_L256_Ltau200 holstein_honeycomb_L8_Ltau80; do
  timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-proc-scan > gpurun_out/r04_quick_$wl.json 2>/dev/null
  python - $wl <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04_quick_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'sweeps/s', round(d['value'],1), 'iters', round(d['avg_cg_iters'],1), 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
PY
done
