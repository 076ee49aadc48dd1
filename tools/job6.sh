#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "96 12 2" "96 12 3" "96 12 4" "96 12 6" "96 6 2" "96 6 3" "192 12 3" "48 6 3"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --walkers-per-gpu $1 --streams $2 --solve-concurrency $3 --matvec-reps 50 --steps 6 --warmup 2 > gpurun_out/scan_$1_$2_$3.json 2>/dev/null
  python - "$1 $2 $3" gpurun_out/scan_$1_$2_$3.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print(sys.argv[1], round(d['value'],1), round(d['config']['avg_cg_iters'],2), round(d['roofline']['avg_launch_us'],1))
PY
done
