#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/r02_t11.log 2>&1; tail -3 gpurun_out/r02_t11.log
