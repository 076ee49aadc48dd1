#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_t7.log 2>&1; tail -8 gpurun_out/r02_t7.log
timeout -k 10 300 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations2.json 2> gpurun_out/device_cases.err; echo pins rc=$?
