#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/solo_w1full
rm -rf $out
cd $GRAFT_REPO_ROOT
SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/one_stream.py 1 > gpurun_out/solo_w1full.log 2>&1
python3 profiles/summarize.py $out 40 | cut -c1-160
tail -1 gpurun_out/solo_w1full.log
rm -rf $out
