#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -4 gpurun_out/r04_gpu_suite.log
for i in 1 2; do SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 python tools/one_stream.py 1 | tail -1; done
SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 python tools/one_stream.py 16 | tail -1
SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r04_1walker 1; tail -22 gpurun_out/gap_r04_1walker.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-proc-scan 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value'],1), 'single', d['single_walker'], [round(x['sweeps_per_s'],1) for x in d['one_stream']])"
