#!/bin/bash
# round 4: the default bench three times on one box (how far one number moves run to run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout -k 10 400 python bench.py --no-proc-scan > gpurun_out/r04_repeat_$rep.json 2> gpurun_out/r04_repeat_$rep.err || exit 1
  python - $rep <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r04_repeat_{sys.argv[1]}.json').read().strip().splitlines()[-1]); r=d['roofline']
print('run', sys.argv[1], 'sweeps/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'MtM in-bench us', round(r['avg_launch_us'],1), 'frac', round(r['frac'],3), 'isolated us', round(r['isolated']['avg_launch_us'],2), 'single walker', round(d['single_walker']['sweeps_per_s'],1), 'cpu', round(d['cpu_baseline']['value'],2), 'async', d['config']['hmc_async'])
PY
done | tee gpurun_out/r04_bench_repeat.txt
