#!/bin/bash
# tfft_rb_kernel at its shipped rule (fewer than 32 systems per launch): full GPU suite, device iteration pins regenerated and diffed, bench records of the three lattices it touches
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -8 gpurun_out/r04_gpu_suite.log
timeout -k 10 600 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.new.json 2> gpurun_out/device_cases.err; echo pins rc=$?
python - <<'PY'
import json
old=json.load(open('tests/golden/device_cg_iterations.json')); new=json.load(open('gpurun_out/device_cg_iterations.new.json'))
for k in sorted(new['device']):
    a,b=old['device'].get(k),new['device'][k]
    if a!=b:
        d=[y-x for x,y in zip(a,b)] if a else None
        print('moved', k, 'deltas', d)
PY
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  t0=$SECONDS
  timeout -k 10 500 python bench.py --workload $wl --steps 6 --warmup 2 --no-proc-scan > gpurun_out/r04_bench_$wl.json 2> gpurun_out/r04_bench_$wl.err; echo $wl rc=$? wall $((SECONDS-t0)) s
  python - $wl <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r04_bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['cpu_baseline']
print(sys.argv[1], 'sweeps/s', round(d['value'],1), 'shape', d['config']['walkers_per_gpu'], 'x', d['config']['streams_per_gpu'], d['config'].get('tfft_kernel'), 'iters', round(d['avg_cg_iters'],1), 'cpu', round(c['value'],2), round(c['avg_cg_iters'],1), 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
print('   roofline', r['kernel'][:40], 'frac', round(r['frac'],3), 'us', round(r['avg_launch_us'],1))
PY
done
