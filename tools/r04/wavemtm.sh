#!/bin/bash
# round 4: the one-wavefront-per-run MᵀM kernel — its tests, the parity files, the stream / bench-shape tests, solo profiles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wave_mtm.py -m gpu -x -q > gpurun_out/r04_wavemtm_tests.log 2>&1; rc=$?; echo wavemtm rc=$rc; tail -25 gpurun_out/r04_wavemtm_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_stream_mtm.py tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_sweep_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r04_wavemtm_parity.log 2>&1; echo parity rc=$?; tail -8 gpurun_out/r04_wavemtm_parity.log
for spec in "bssh 16 bssh_chain_L256_Ltau200" "ossh 16 ossh_square_L12_Ltau100" "hc16 16 holstein_honeycomb_L16_Ltau128" "hc16_w1 1 holstein_honeycomb_L16_Ltau128" "hc8 16 holstein_honeycomb_L8_Ltau80"; do
  set -- $spec; bash tools/solo_profile.sh r04_$1 $2 $3 || exit 1; head -8 gpurun_out/solo_r04_$1.txt | cut -c1-150 | grep -v "^#\|^kernel"
done
