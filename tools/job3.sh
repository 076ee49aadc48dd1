#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "96 6 3" "96 6 4" "96 6 6" "128 8 4" "96 4 4" "96 3 3" "128 4 4"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --hmc device --walkers-per-gpu $1 --streams $2 --solve-concurrency $3 --matvec-reps 50 --steps 6 --warmup 2 > gpurun_out/scan_$1_$2_$3.json 2>/dev/null
  python - "$1 $2 $3" gpurun_out/scan_$1_$2_$3.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], round(d['value'],1), round(d['config']['avg_cg_iters'],2), round(d['roofline']['avg_launch_us'],1))
PY
done
