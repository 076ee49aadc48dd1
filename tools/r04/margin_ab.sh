#!/bin/bash
# round 4: the asynchronous trajectory's margin (early-exit iterations past the hint): repeats and sweeps/s of the timed region, two repetitions
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --timed-only --steps 10 --warmup 2 > gpurun_out/r04_margin_$rep.json 2> gpurun_out/r04_margin_$rep.err || exit 1
  python - $rep <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r04_margin_{sys.argv[1]}.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['config']['hmc_async'])
PY
done
