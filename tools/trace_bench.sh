#!/bin/bash
# kernel timeline of a bench run: tools/trace_bench.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/trace_$tag.log 2>&1
python3 tools/trace_overlap.py $out ${TRACE_FROM:-0.55} ${TRACE_TO:-1.0} > gpurun_out/trace_$tag.txt
grep -o '"value": [0-9.]*' gpurun_out/trace_$tag.log | head -1 >> gpurun_out/trace_$tag.txt
rm -rf $out
