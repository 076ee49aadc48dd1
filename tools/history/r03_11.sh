#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -m gpu -q -x -k "kpm or pcg or precond" > gpurun_out/r03_t11.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_t11.log
bash tools/solo_profile.sh r03_hc16_dpp 16 && bash tools/solo_profile.sh r03_1walker_dpp 1
for t in r03_hc16_dpp r03_1walker_dpp; do echo "== $t"; head -9 gpurun_out/solo_$t.txt; tail -1 gpurun_out/solo_$t.txt; done
