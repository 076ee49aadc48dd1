#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/r04/lanczos.sh | tee gpurun_out/r04_lanczos_steps_b.txt
timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -m gpu -x -q 2>&1 | tail -3
for nw in 1 16; do SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw | tail -1; done
SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 64 holstein_honeycomb_L8_Ltau80 | tail -1
