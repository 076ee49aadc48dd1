#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_wave_mtm.py -m gpu -x -q 2>&1 | tail -3
python tools/wave_scan.py holstein_honeycomb_L16_Ltau128 1,4,16,64,128 0,-1,2,4,8,16 | cut -c1-330
python tools/wave_scan.py ossh_square_L12_Ltau100 16,128 0,-1,2,4,8 | cut -c1-300
python tools/wave_scan.py bssh_chain_L256_Ltau200_alpha0p2 16,128 0,-1,2,4,8,16 | cut -c1-300
python tools/wave_scan.py holstein_honeycomb_L8_Ltau80 16,128 0,-1,2,4,8 | cut -c1-300
