#!/bin/bash
cd $GRAFT_REPO_ROOT
for nw in 32 64; do for sp in 1 2 3 4; do echo "nw=$nw parts=$sp: $(SMOQY_SPLIT=$sp SMOQY_EFA=1 timeout -k 10 200 python tools/one_stream.py $nw 2>&1 | tail -1)"; done; done
