#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_efa.py -m gpu -x -q > gpurun_out/r02_t4.log 2>&1; tail -15 gpurun_out/r02_t4.log
timeout -k 10 300 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.json 2> gpurun_out/device_cases.err; echo pins rc=$?
bash tools/gap_probe.sh w1 1 && bash tools/gap_probe.sh w16 16
echo done
