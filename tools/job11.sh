#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_irregular.py tests/test_gpu_edge_cases.py tests/test_gpu_api.py tests/test_gpu_large_lattice.py tests/test_golden.py -m gpu -x -q > gpurun_out/r02_t8.log 2>&1; tail -6 gpurun_out/r02_t8.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_shape.py -m gpu -x -q -k asym > gpurun_out/r02_t9.log 2>&1; tail -12 gpurun_out/r02_t9.log
bash tools/solo_profile.sh r02_hc16_asym2 16 holstein_honeycomb_L16_Ltau128 asym; head -8 gpurun_out/solo_r02_hc16_asym2.txt | cut -c1-150; tail -1 gpurun_out/solo_r02_hc16_asym2.txt
