#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_efa.py tests/test_gpu_team.py tests/test_gpu_complex_T.py -m gpu -q -x 2>&1 | tail -3
cat /proc/loadavg
for rep in 1 2; do for nw in 1 16 64; do echo "nw=$nw: $(timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; done; done
echo "hc8 nw=1: $(timeout -k 10 120 python tools/history/one_stream_iters.py 1 holstein_honeycomb_L8_Ltau80 2>&1 | tail -1)"
for rep in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only > gpurun_out/r03_b58.json 2>gpurun_out/r03_b58.err; python -c "import json; d=json.load(open('gpurun_out/r03_b58.json')); print('bench', round(d['value'],1), round(d['roofline']['avg_launch_us'],1), d['steps'])"; done
