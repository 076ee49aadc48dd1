#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/solo_profile.sh r03_hc16 16 && bash tools/solo_profile.sh r03_1walker 1 && SMOQY_CHEB_WL0=0 bash tools/solo_profile.sh r03_hc16_wl0off 16 && SMOQY_CHEB_WL0=0 bash tools/solo_profile.sh r03_1walker_wl0off 1
for t in r03_hc16 r03_hc16_wl0off r03_1walker r03_1walker_wl0off; do echo "== $t"; head -9 gpurun_out/solo_$t.txt; tail -1 gpurun_out/solo_$t.txt; done
