#!/bin/bash
# Solo (single stream) per-kernel durations of one sweep: tools/solo_profile.sh <tag> [walkers]
# Writes gpurun_out/solo_<tag>.txt
tag=$1; nw=${2:-16}  # optional third / fourth argument: workload name, sym|asym
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/solo_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/one_stream.py $nw $3 $4 > gpurun_out/solo_$tag.log 2>&1
python3 profiles/summarize.py $out 12 > gpurun_out/solo_$tag.txt
tail -1 gpurun_out/solo_$tag.log >> gpurun_out/solo_$tag.txt
rm -rf $out
