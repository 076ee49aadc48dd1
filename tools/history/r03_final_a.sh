#!/bin/bash
# closing sequence, step 1: regenerate the device iteration pins at HEAD, then the whole GPU suite against them
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.json 2> gpurun_out/device_cases.err; echo "pins rc=$?"
python -c "import json; d=json.load(open('gpurun_out/device_cg_iterations.json')); print(len(d['device']), 'cases')" && cp gpurun_out/device_cg_iterations.json tests/golden/device_cg_iterations.json
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r03_gpu_suite.log
