#!/bin/bash
# NOTE: historical — the paired-frequency kernel (SMOQY_CHEB_PAIR) was measured with the build of commit "real-vector ldiv! …" and then removed;
# the switch no longer exists.  Kept as the record of how profiles/r02_solo_kernel_stats_cheb_pair_on/off.txt were taken.
# A/B of the paired Chebyshev kernel (SMOQY_CHEB_PAIR): parity tests, solo profiles at 16 and 1 walkers, default bench
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02_t2.log 2>&1
tail -3 gpurun_out/r02_t2.log
bash tools/solo_profile.sh pair1 16 && SMOQY_CHEB_PAIR=0 bash tools/solo_profile.sh pair0 16 && bash tools/solo_profile.sh pair1_w1 1 && SMOQY_CHEB_PAIR=0 bash tools/solo_profile.sh pair0_w1 1 \
 && timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_pair1.json 2>/dev/null && SMOQY_CHEB_PAIR=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_pair0.json 2>/dev/null
echo rc=$?
