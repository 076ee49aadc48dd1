#!/bin/bash
# per-kernel totals of a default bench run restricted to warm-up + timed region (--timed-only: no isolated / scan launches dilute
# the averages), with its own JSON line (roofline.avg_launch_us from the device clock) kept next to it: tools/profile_bench.sh <tag> [bench args]
# bench.py --timed-only writes gpurun_out/bench_maps_rank0.txt (its /proc/self/maps once every library is loaded); if the profiled run
# aborts, that map and the tail of the log are kept as gpurun_out/<tag>_abort_maps.txt / _abort_log.txt so the frames can be attributed —
# the run is NOT repeated here.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --steps 2 --warmup 1 --timed-only "$@" > gpurun_out/prof_$tag.log 2>&1
rc=$?
if [ $rc -ne 0 ]; then
  cp gpurun_out/bench_maps_rank0.txt gpurun_out/${tag}_abort_maps.txt 2>/dev/null
  tail -80 gpurun_out/prof_$tag.log > gpurun_out/${tag}_abort_log.txt
  echo "profiled bench run ended with rc=$rc: maps and log tail kept as gpurun_out/${tag}_abort_*"
  exit $rc
fi
python3 profiles/summarize.py $out 30 > gpurun_out/${tag}_bench_kernel_stats.txt
cp $(ls $out/*kernel_stats.csv $out/*/*kernel_stats.csv 2>/dev/null | head -1) gpurun_out/${tag}_bench_kernel_stats.csv
rm -rf $out
grep '^{' gpurun_out/prof_$tag.log | tail -1 > gpurun_out/${tag}_bench_profiled_run.json
