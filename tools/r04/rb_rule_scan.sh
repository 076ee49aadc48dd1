#!/bin/bash
# where tfft_rb_kernel stops paying: one stream at 32 and 64 walkers, and the eight-stream bench at 32 systems per launch, rule (SMOQY_TFFT_EDGE=2: off from 32 systems) against forced (=3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_golden.py tests/test_gpu_edge_cases.py -m gpu -q -k "register_blocked or pinned" > gpurun_out/r04_rb_tests2.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r04_rb_tests2.log
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  for nw in 32 64; do
    for e in 2 3 2 3; do
      echo "$wl one stream $nw walkers SMOQY_TFFT_EDGE=$e $(SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_TFFT_EDGE=$e timeout -k 10 200 python tools/one_stream.py $nw $wl 2> /dev/null | tail -1)"
    done
  done
done | tee gpurun_out/r04_rb_rule_scan.txt
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100; do
  for e in 2 3 2 3; do
    SMOQY_TFFT_EDGE=$e timeout -k 10 300 python bench.py --workload $wl --walkers-per-gpu 256 --streams 8 --timed-only --steps 6 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '256 x 8 SMOQY_TFFT_EDGE=$e', round(d['value'],1), 'sweeps/s', d['config'].get('tfft_kernel'))" || exit 1
  done
done | tee -a gpurun_out/r04_rb_rule_scan.txt
