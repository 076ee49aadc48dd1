#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
s=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r03_bench_output.json 2> gpurun_out/r03_bench_output.err; echo bench rc=$? in $(( $(date +%s) - s )) s
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_output.json'))
print(d['value'], d['single_walker']['sweeps_per_s'])
for k in ('team_threads_scan','team_procs_scan'):
    for p in d[k]['points']: print(k, {a:(round(b,1) if isinstance(b,float) else b) for a,b in p.items()})
PY
