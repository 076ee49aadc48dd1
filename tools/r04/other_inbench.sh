#!/bin/bash
# round 4: per-kernel totals INSIDE the bench (timed region) for the other lattices' default shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_bench.sh r04_ossh --workload ossh_square_L12_Ltau100 && head -14 gpurun_out/r04_ossh_bench_kernel_stats.txt | cut -c1-170 &&
bash tools/profile_bench.sh r04_bssh --workload bssh_chain_L256_Ltau200 && head -14 gpurun_out/r04_bssh_bench_kernel_stats.txt | cut -c1-170 &&
bash tools/profile_bench.sh r04_hc8 --workload holstein_honeycomb_L8_Ltau80 && head -14 gpurun_out/r04_hc8_bench_kernel_stats.txt | cut -c1-170
