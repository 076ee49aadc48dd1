#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r03_gpu_suite3.log 2>&1; echo "suite rc=$?"; tail -25 gpurun_out/r03_gpu_suite3.log
for nw in 1 16; do SMOQY_EFA=1 SMOQY_SPLIT=0 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1; done
SMOQY_EFA=1 bash tools/gap_probe.sh r03_1walker_efa_b 1; cat gpurun_out/gap_r03_1walker_efa_b.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b3_q4.json 2>gpurun_out/r03_b3_q4.err; python -c "import json; d=json.load(open('gpurun_out/r03_b3_q4.json')); print('q4', d['value'], d['roofline']['avg_launch_us'])"
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python bench.py --no-cpu-baseline --no-proc-scan --timed-only --steps 6 --warmup 2 > gpurun_out/r03_b3_q8.json 2>gpurun_out/r03_b3_q8.err; python -c "import json; d=json.load(open('gpurun_out/r03_b3_q8.json')); print('q8', d['value'], d['roofline']['avg_launch_us'])"
