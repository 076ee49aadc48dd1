#!/bin/bash
# round 4: the wavefront-PAIR form of the MᵀM kernel (fdm_wave2_kernel): its tests, then the scan against form 1 and the workgroup kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_wave_mtm.py -m gpu -x -q > gpurun_out/r04_pair_tests.log 2>&1; rc=$?; echo pair tests rc=$rc; tail -25 gpurun_out/r04_pair_tests.log
[ $rc -ne 0 ] && exit $rc
python tools/wave_scan.py holstein_honeycomb_L16_Ltau128 16,64,128 0,2,4,8,16 0 1,2 2>&1 | tee gpurun_out/r04_pair_scan.txt
python tools/wave_scan.py ossh_square_L12_Ltau100 16,128 0,2,4,10 0 1,2 2>&1 | tee -a gpurun_out/r04_pair_scan.txt
python tools/wave_scan.py bssh_chain_L256_Ltau200 16,128 0,2,4,8 0 1,2 2>&1 | tee -a gpurun_out/r04_pair_scan.txt
python tools/wave_scan.py holstein_honeycomb_L8_Ltau80 16,64 0,2,4 0 1,2 2>&1 | tee -a gpurun_out/r04_pair_scan.txt
