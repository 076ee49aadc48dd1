#!/bin/bash
# round 4, evidence for BASELINE configs 2, 3, 5 (and the smallest lattice): walkers x streams scan (timed region only), PMC traffic of the
# fused MtM and of the whole iteration at each lattice, solo profiles; the full bench lines follow in tools/r04/configs_bench.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r04_shape_scan.txt
for wl in ${WORKLOADS:-holstein_honeycomb_L8_Ltau80 ossh_square_L12_Ltau100 bssh_chain_L256_Ltau200}; do
  for cfg in "128 8" "256 8" "512 8" "256 4" "128 4"; do
    read -r w s <<< "$cfg"
    v=$(timeout -k 10 300 python bench.py --workload $wl --timed-only --steps ${STEPS:-3} --warmup 1 --walkers-per-gpu $w --streams $s --no-mtm-sampling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['avg_cg_iters'],1))")
    echo "$wl walkers $w streams $s: $v" | tee -a gpurun_out/r04_shape_scan.txt
  done
done
for spec in "hc8 holstein_honeycomb_L8_Ltau80" "ossh ossh_square_L12_Ltau100" "bssh bssh_chain_L256_Ltau200"; do
  set -- $spec
  bash tools/pmc_traffic.sh 16 $2 $1 > /dev/null && echo pmc traffic $1 ok
  PMC_WORKLOAD=$2 PMC_TAG=$1 bash tools/pmc_iteration.sh | cut -c1-160
done
