#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_efa.py -m gpu -q -x 2>&1 | tail -5
cat /proc/loadavg
for rep in 1 2; do for pf in 1 0; do for nw in 1 16; do echo "prefetch=$pf nw=$nw: $(SMOQY_PREFETCH=$pf timeout -k 10 120 python tools/history/one_stream_iters.py $nw 2>&1 | tail -1)"; done; done; done
for rep in 1 2 3; do for pf in "" "--no-prefetch"; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only $pf > gpurun_out/r03_b54.json 2>gpurun_out/r03_b54.err; python -c "import json; d=json.load(open('gpurun_out/r03_b54.json')); print('bench [$pf]', round(d['value'],1), round(d['roofline']['avg_launch_us'],1), d['steps'])"; done; done
cat /proc/loadavg
