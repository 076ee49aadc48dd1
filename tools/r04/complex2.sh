#!/bin/bash
# round 4: register-resident operator kernel for complex hoppings: its tests, the real-T pins (arithmetic must not have moved), the scan
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_complex_T.py -m gpu -x -q > gpurun_out/r04_complex_tests.log 2>&1; rc=$?; echo complex rc=$rc; tail -15 gpurun_out/r04_complex_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py tests/test_gpu_kernel_families.py tests/test_gpu_greens.py tests/test_gpu_force.py -m gpu -x -q > gpurun_out/r04_complex_pins.log 2>&1; rc=$?; echo pins rc=$rc; tail -5 gpurun_out/r04_complex_pins.log
[ $rc -ne 0 ] && exit $rc
python tools/complex_scan.py 16 2>&1 | tee gpurun_out/r04_complex_scan_b.txt
python tools/complex_scan.py 1 2>&1 | tee -a gpurun_out/r04_complex_scan_b.txt
