#!/bin/bash
cd $GRAFT_REPO_ROOT
for sb in 8 16; do
  SMOQY_TFFT_SB=$sb timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 > gpurun_out/sb_$sb.json 2>/dev/null
  python - $sb <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/sb_{sys.argv[1]}.json')); print('SB',sys.argv[1], round(d['value'],1), [round(x['sweeps_per_s'],1) for x in d['one_stream']], round(d['roofline']['avg_launch_us'],1))
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 > gpurun_out/sb_def.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/sb_def.json')); print('default', round(d['value'],1), [round(x['sweeps_per_s'],1) for x in d['one_stream']], round(d['roofline']['avg_launch_us'],1))
PY
