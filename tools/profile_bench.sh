#!/bin/bash
# per-kernel totals of one default bench run + the un-profiled JSON line: tools/profile_bench.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
python3 profiles/summarize.py $out 30 > gpurun_out/${tag}_bench_kernel_stats.txt
cp $(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf $out
