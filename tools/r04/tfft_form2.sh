#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in holstein_honeycomb_L8_Ltau80 holstein_honeycomb_L4_Ltau40; do
  for form in two-image in-place two-image in-place two-image in-place; do
    timeout -k 10 300 python bench.py --workload $wl --timed-only --steps 20 --warmup 3 --tfft-form $form 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', '$form', round(d['value'],1), 'sweeps/s', d['config']['tfft_form'])"
  done
done | tee -a gpurun_out/r04_tfft_form_scan.txt
