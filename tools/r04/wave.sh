#!/bin/bash
# round 4: the one-wavefront-per-chain Chebyshev kernel — DPP probe, its tests, the parity files, the pins, solo profiles of the two SSH lattices
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe > gpurun_out/r04_dpp_probe.txt; cat gpurun_out/r04_dpp_probe.txt
timeout -k 10 900 python -m pytest tests/test_gpu_cheb_wave.py -m gpu -x -q > gpurun_out/r04_wave_tests.log 2>&1; echo wave rc=$?; tail -15 gpurun_out/r04_wave_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_golden.py tests/test_gpu_sweep_parity.py -m gpu -q > gpurun_out/r04_wave_parity.log 2>&1; echo parity rc=$?; tail -15 gpurun_out/r04_wave_parity.log
bash tools/solo_profile.sh r04_bssh 16 bssh_chain_L256_Ltau200 && bash tools/solo_profile.sh r04_ossh 16 ossh_square_L12_Ltau100 && head -14 gpurun_out/solo_r04_bssh.txt gpurun_out/solo_r04_ossh.txt
ls gpurun_out | grep r04
