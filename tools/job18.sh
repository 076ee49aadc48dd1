#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "python 3" "library 2" "library 3" "library 4" "library 6" "python 3" "library 3"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --timed-only --steps 8 --gate $1 --solve-concurrency $2 > gpurun_out/g_$1_$2.json 2>/dev/null
  python - "$1 $2" gpurun_out/g_$1_$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print('gate',sys.argv[1], round(d['value'],1), round(d['roofline']['avg_launch_us'],1))
PY
done
