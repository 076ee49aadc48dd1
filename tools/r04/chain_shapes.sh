#!/bin/bash
# bond-SSH chain with tfft_rb_kernel: walkers x streams shapes around the committed best (128 x 8), timed region only
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for shape in "128 8" "192 12" "128 16" "256 16" "96 6" "128 8" "192 12" "128 16"; do
  set -- $shape
  timeout -k 10 300 python bench.py --workload bssh_chain_L256_Ltau200 --walkers-per-gpu $1 --streams $2 --timed-only --steps 6 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bssh_chain_L256_Ltau200 walkers $1 streams $2:', round(d['value'],1), 'sweeps/s', d['config'].get('tfft_kernel')[:30])" || echo "shape $1 x $2 failed"
done | tee gpurun_out/r04_chain_shapes_rb.txt
