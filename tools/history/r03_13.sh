#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_complex_T.py tests/test_gpu_force.py tests/test_gpu_phonon_fields.py tests/test_gpu_c_abi.py -m gpu -q > gpurun_out/r03_t13.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r03_t13.log
