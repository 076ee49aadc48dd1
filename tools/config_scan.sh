#!/bin/bash
# interleaved comparison of walker/stream/split/gate configurations of bench.py (host load printed: the result depends on it)
cd $GRAFT_REPO_ROOT
cat /proc/loadavg
for rep in 1 2; do
for cfg in "96 6 1 3" "96 3 2 2" "96 3 2 3" "128 4 2 3" "96 4 2 3"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --timed-only --steps 6 --walkers-per-gpu $1 --streams $2 --cg-split $3 --solve-concurrency $4 --no-mtm-sampling > gpurun_out/s_$1_$2_$3_$4.json 2>/dev/null
  python - "$cfg" gpurun_out/s_$1_$2_$3_$4.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print('walkers streams split gate', sys.argv[1], round(d['value'],1), flush=True)
PY
done
cat /proc/loadavg
done
