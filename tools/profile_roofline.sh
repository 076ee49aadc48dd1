#!/bin/bash
# rocprofv3 pass over the isolated roofline leg of bench.py: tools/profile_roofline.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/proofline_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --roofline-only > gpurun_out/${tag}_bench_roofline_only.json 2> gpurun_out/proofline_$tag.log
python3 profiles/summarize.py $out 6 > gpurun_out/${tag}_bench_roofline_kernel_stats.txt
rm -rf $out
