#!/bin/bash
# last call of round 4: the GPU suite and smoke() at the final HEAD, and the per-kernel totals of the timed region of the bench on the three other lattices
# (what profiles/r04_bench_kernel_stats.txt is for the headline) -> gpurun_out/r04_<tag>_bench_kernel_stats.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo smoke rc=$?; tail -2 gpurun_out/r04_smoke.log
bash tools/profile_bench.sh r04_bssh --workload bssh_chain_L256_Ltau200 && echo prof bssh ok
bash tools/profile_bench.sh r04_ossh --workload ossh_square_L12_Ltau100 && echo prof ossh ok
bash tools/profile_bench.sh r04_hc8 --workload holstein_honeycomb_L8_Ltau80 --steps 10 && echo prof hc8 ok
for t in bssh ossh hc8; do head -8 gpurun_out/r04_${t}_bench_kernel_stats.txt | cut -c1-160; done
