#!/bin/bash
# tile width of the in-place / two-image tau-FFT at 64 systems per launch on four streams: 16 sites (default at Ltau = 80, 100) against 8 (twice the workgroups, 128-byte rows)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  steps=6; [ $wl = holstein_honeycomb_L8_Ltau80 ] && steps=40
  for sb in 16 8 16 8; do
    SMOQY_TFFT_SB=$sb timeout -k 10 300 python bench.py --workload $wl --timed-only --steps $steps --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl SMOQY_TFFT_SB=$sb', round(d['value'],1), 'sweeps/s', d['config'].get('tfft_kernel')[:40])" || exit 1
  done
done | tee gpurun_out/r04_tfft_sb_scan.txt
