#!/bin/bash
# round 4: lanczos_wave_kernel: the full GPU suite (pins included), then duration against the number of steps, then one-stream sweeps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_lanczos_suite.log 2>&1; rc=$?; echo suite rc=$rc; tail -12 gpurun_out/r04_lanczos_suite.log
bash tools/r04/lanczos.sh | tee gpurun_out/r04_lanczos_steps.txt
for nw in 1 16; do SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw | tail -1; SMOQY_LANCZOS_WAVE=0 SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw | tail -1; done
SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 64 holstein_honeycomb_L8_Ltau80 | tail -1; SMOQY_LANCZOS_WAVE=0 SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 64 holstein_honeycomb_L8_Ltau80 | tail -1
