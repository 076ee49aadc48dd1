#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_efa.py tests/test_gpu_sweep_parity.py tests/test_gpu_team.py -m gpu -x -q > gpurun_out/r04_async_tests.log 2>&1; echo tests rc=$?; tail -12 gpurun_out/r04_async_tests.log
for rep in 1 2; do for a in 0 1; do for nw in 1 16; do
  r=$(SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_ASYNC=$a python tools/one_stream.py $nw | tail -1); echo "async=$a nw=$nw: $r"
done; done; done
for rep in 1 2; do for a in 0 1; do
 v=$(SMOQY_BENCH_ASYNC=$a timeout -k 10 300 python bench.py --timed-only --steps 6 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
 echo "bench async=$a: $v"
done; done
