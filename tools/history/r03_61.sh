#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03_gpu_suite.log
for nw in 32 64; do echo "auto nw=$nw: $(timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; done
timeout -k 10 300 python tools/team_scan.py 64 4 2x32 2>&1 | grep -o '"threads": [0-9]*\|"teams": [0-9]*\|native_threads_sweeps_per_s": [0-9.]*' | tr '\n' ' '
timeout -k 10 300 python bench.py --no-cpu-baseline --no-proc-scan --roofline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print([(b['batch'],round(b['us'],1)) for b in r['batch_scan']]); print(r['hbm_resident_point'])"
