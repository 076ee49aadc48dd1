#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sb in 8 4; do for nw in 1 2 4; do echo "SB=$sb nw=$nw: $(SMOQY_TFFT_SB=$sb SMOQY_EFA=1 timeout -k 10 120 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done; done
SMOQY_TFFT_SB=4 bash tools/solo_profile.sh r03_sb4_w1 1 > /dev/null 2>&1; head -8 gpurun_out/solo_r03_sb4_w1.txt | cut -c1-150
