#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for gr in 0 1; do
  r=$(SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_ASYNC=1 SMOQY_GRAPH=$gr python tools/one_stream.py 1 | tail -1); echo "async=1 graph=$gr nw=1: $r"
done; done
SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_ASYNC=1 bash tools/gap_probe.sh r04_1walker 1; tail -8 gpurun_out/gap_r04_1walker.txt
