#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SMOQY_TFFT_PREFETCH=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -m gpu -q -x > gpurun_out/r03_t45.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_t45.log
for rep in 1 2; do for pf in 0 1; do for nw in 1 16; do echo "prefetch=$pf nw=$nw: $(SMOQY_TFFT_PREFETCH=$pf SMOQY_EFA=1 SMOQY_SPLIT=1 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done; done
SMOQY_TFFT_PREFETCH=1 bash tools/solo_profile.sh r03_pf1 16 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_pf1.txt | cut -c1-150
SMOQY_TFFT_PREFETCH=1 bash tools/solo_profile.sh r03_pf1_w1 1 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_pf1_w1.txt | cut -c1-150
