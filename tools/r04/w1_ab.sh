#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
 for v in "0 0" "1 0" "0 1"; do set -- $v
  r=$(SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_GRAPH=$1 SMOQY_POLL=$2 python tools/one_stream.py 1 | tail -1)
  echo "graph=$1 poll=$2: $r"
 done
done
