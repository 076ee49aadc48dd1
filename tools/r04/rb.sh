#!/bin/bash
# register-blocked tau-FFT for Ltau = 80 / 100 / 200 (tfft_rb_kernel): focused tests, solo kernel durations, interleaved A/B of the timed region
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_edge_cases.py -m gpu -q -x -k "tau_fft or fft" > gpurun_out/r04_rb_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -4 gpurun_out/r04_rb_tests.log
[ $rc -ne 0 ] && exit $rc
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  for edge in 1 2 1 2; do
    SMOQY_TFFT_EDGE=$edge timeout -k 10 300 python bench.py --workload $wl --timed-only --steps 10 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'SMOQY_TFFT_EDGE=$edge', round(d['value'],1), 'sweeps/s', d['config'].get('tfft_form'), d['config'].get('avg_cg_iters'))" || exit 1
  done
done | tee gpurun_out/r04_rb_bench_ab.txt
bash tools/solo_profile.sh r04_bssh_rb 16 bssh_chain_L256_Ltau200 && bash tools/solo_profile.sh r04_ossh_rb 16 ossh_square_L12_Ltau100 && bash tools/solo_profile.sh r04_hc8_rb 16 holstein_honeycomb_L8_Ltau80
head -9 gpurun_out/solo_r04_bssh_rb.txt; head -9 gpurun_out/solo_r04_ossh_rb.txt; head -9 gpurun_out/solo_r04_hc8_rb.txt
