#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "4 3" "8 3" "8 4" "8 6" "12 6"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --timed-only --steps 8 --solve-concurrency $2 > gpurun_out/q_$1_$2.json 2>/dev/null
  python - "$1 $2" gpurun_out/q_$1_$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print('queues,gate',sys.argv[1], round(d['value'],1), round(d['roofline']['avg_launch_us'],1))
PY
done
GPU_MAX_HW_QUEUES=8 bash tools/trace_bench.sh r02q8 --timed-only --steps 2 --solve-concurrency 4; cat gpurun_out/trace_r02q8.txt | head -8
