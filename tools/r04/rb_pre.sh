#!/bin/bash
# (historical: the kernels of this experiment were removed again, docs/DESIGN_LOG.md R4.13) epilogue operands prefetched (PRE; SMOQY_TFFT_EDGE=4 switched it off): bit-identity tests, pins, one- and eight-walker sweeps with the prefetch off and on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_edge_cases.py tests/test_gpu_efa.py -m gpu -q -k "prefetched or register_blocked or tau_fft or pinned or efa" > gpurun_out/r04_rb_pre_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 gpurun_out/r04_rb_pre_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in holstein_honeycomb_L16_Ltau128 bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  for nw in 1 8; do
    [ $nw -eq 8 ] && [ $wl != holstein_honeycomb_L16_Ltau128 ] && continue
    for e in 4 2 4 2; do
      echo "$wl one stream $nw walkers SMOQY_TFFT_EDGE=$e $(SMOQY_SWEEPS=8 SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_TFFT_EDGE=$e timeout -k 10 300 python tools/one_stream.py $nw $wl 2> /dev/null | tail -1)"
    done
  done
done | tee gpurun_out/r04_rb_pre_scan.txt
bash tools/solo_profile.sh r04_hc16_w1_pre 1 && head -9 gpurun_out/solo_r04_hc16_w1_pre.txt
# honeycomb L = 8 at its bench shape (4 x 64), long enough to tell: two-image (SMOQY_TFFT_EDGE=1) against tfft_rb_kernel (the one-stream rule lets it run at 64 systems when nobody asked for the in-place form)
for e in 1 2 1 2 1 2; do
  SMOQY_TFFT_EDGE=$e timeout -k 10 300 python bench.py --workload holstein_honeycomb_L8_Ltau80 --timed-only --steps 60 --warmup 5 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('holstein_honeycomb_L8_Ltau80 4 x 64 SMOQY_TFFT_EDGE=$e', round(d['value'],1), 'sweeps/s', d['config'].get('tfft_kernel')[:40])" || exit 1
done | tee -a gpurun_out/r04_rb_pre_scan.txt
