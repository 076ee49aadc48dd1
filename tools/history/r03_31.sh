#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nw in 32 64 128; do for slim in 0 1; do echo "nw=$nw slim=$slim: $(SMOQY_TFFT_SLIM=$slim SMOQY_EFA=1 timeout -k 10 200 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done
SMOQY_TFFT_SLIM=1 SMOQY_EFA=1 bash tools/solo_profile.sh r03_hc64_slim 64 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_hc64_slim.txt | cut -c1-160
