#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/r04/lanczos.sh | tee gpurun_out/r04_lanczos_steps.txt
NW=16 bash tools/r04/lanczos.sh | tee -a gpurun_out/r04_lanczos_steps.txt
for nw in 1 16; do SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw | tail -1; SMOQY_LANCZOS_WAVE=0 SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw | tail -1; done
SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 64 holstein_honeycomb_L8_Ltau80 | tail -1; SMOQY_LANCZOS_WAVE=0 SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py 64 holstein_honeycomb_L8_Ltau80 | tail -1
