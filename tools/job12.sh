#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r02_t10.log 2>&1; tail -5 gpurun_out/r02_t10.log
bash tools/solo_profile.sh stage1 16; head -8 gpurun_out/solo_stage1.txt | cut -c1-150; tail -1 gpurun_out/solo_stage1.txt
bash tools/solo_profile.sh stage1_w1 1; head -8 gpurun_out/solo_stage1_w1.txt | cut -c1-150; tail -1 gpurun_out/solo_stage1_w1.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_stage1.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r02_bench_stage1.json'));print(d['value'],[round(x['sweeps_per_s'],1) for x in d['one_stream']], d['roofline']['avg_launch_us'], d['roofline']['isolated']['avg_launch_us'])"
