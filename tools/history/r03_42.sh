#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for xm in 0 -1; do for cfg in "8 1" "16 0" "16 1" "32 0"; do set -- $cfg; if [ $xm = -1 ]; then unset SMOQY_XCD_MAP; else export SMOQY_XCD_MAP=$xm; fi; echo "xcd_map=$xm nw=$1 split=$2: $(SMOQY_EFA=1 SMOQY_SPLIT=$2 timeout -k 10 120 python tools/one_stream.py $1 2>&1 | tail -1)"; done; done; done
unset SMOQY_XCD_MAP
for wl in holstein_honeycomb_L8_Ltau80 bssh_chain_L256_Ltau200_alpha0p2; do for xm in 0 -1; do if [ $xm = -1 ]; then unset SMOQY_XCD_MAP; else export SMOQY_XCD_MAP=$xm; fi; echo "$wl xcd_map=$xm nw=16: $(SMOQY_EFA=1 SMOQY_SPLIT=1 timeout -k 10 120 python tools/one_stream.py 16 $wl 2>&1 | tail -1)"; done; done
