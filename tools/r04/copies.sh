#!/bin/bash
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/copies; rm -rf $out
cd $GRAFT_REPO_ROOT
SMOQY_EFA=1 SMOQY_PREFETCH=1 rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 tools/one_stream.py ${1:-16} > gpurun_out/copies.log 2>&1
python3 tools/copy_hist.py $out | tee gpurun_out/r04_copy_hist_${1:-16}.txt
rm -rf $out
