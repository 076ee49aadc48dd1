#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_irregular.py -m gpu -x -q > gpurun_out/r02_t6.log 2>&1; tail -8 gpurun_out/r02_t6.log
bash tools/solo_profile.sh split1 16 && SMOQY_CHEB_SPLIT=0 bash tools/solo_profile.sh split0 16 && bash tools/solo_profile.sh split1_w1 1 && SMOQY_CHEB_SPLIT=0 bash tools/solo_profile.sh split0_w1 1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_split1.json 2>/dev/null; SMOQY_CHEB_SPLIT=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_split0.json 2>/dev/null
echo done
