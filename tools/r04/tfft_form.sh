#!/bin/bash
# round 4: tau-FFT form (two-image / in-place) on the lattices whose Ltau has a factor 5: timed region only, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  for form in two-image in-place two-image in-place; do
    timeout -k 10 300 python bench.py --workload $wl --timed-only --steps ${STEPS:-3} --warmup 1 --tfft-form $form 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$form', round(d['value'],1), 'sweeps/s', d['config']['tfft_form'])"
  done
done | tee gpurun_out/r04_tfft_form_scan.txt
