#!/bin/bash
# round 4: what the Lanczos kernel's time is made of: duration against the number of steps
cd /tmp && export TMPDIR=/tmp
for n in 2 10 20 40; do
  out=$GRAFT_REPO_ROOT/gpurun_out/lz_$n; rm -rf $out
  ( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tools/lanczos_probe.py $n ${NW:-1} > gpurun_out/lz_$n.log 2>&1 )
  echo "n=$n $(grep lanczos $(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1) | sed "s/.*)\",//")"
  rm -rf $out
done
