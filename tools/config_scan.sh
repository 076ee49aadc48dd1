#!/bin/bash
# interleaved comparison of walker/stream/split/gate configurations of bench.py (host load printed: the result depends on it)
# usage: tools/config_scan.sh ["walkers streams split gate" ...]
cd $GRAFT_REPO_ROOT
[ $# -eq 0 ] && set -- "96 6 1 3" "96 3 2 2" "96 3 2 3" "128 4 2 3" "96 4 2 3"
cat /proc/loadavg
for rep in 1 2; do
for cfg in "$@"; do
  read -r w s sp g <<< "$cfg"
  timeout -k 10 300 python bench.py --timed-only --steps 6 --walkers-per-gpu $w --streams $s --cg-split $sp --solve-concurrency $g --no-mtm-sampling > gpurun_out/s_${w}_${s}_${sp}_${g}.json 2>/dev/null
  python - "$cfg" gpurun_out/s_${w}_${s}_${sp}_${g}.json <<'PY'
import json,sys
d=json.load(open(sys.argv[2])); print('walkers streams split gate', sys.argv[1], round(d['value'],1), flush=True)
PY
done
cat /proc/loadavg
done
