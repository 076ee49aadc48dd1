#!/bin/bash
# kernel-boundary gaps of a one-stream run: tools/gap_probe.sh <tag> [walkers] [workload]
tag=$1; nw=${2:-1}; wl=${3:-holstein_honeycomb_L16_Ltau128}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/gap_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 tools/one_stream.py $nw $wl > gpurun_out/gap_$tag.log 2>&1
python3 tools/gap_probe.py $out > gpurun_out/gap_$tag.txt
tail -1 gpurun_out/gap_$tag.log >> gpurun_out/gap_$tag.txt
rm -rf $out
