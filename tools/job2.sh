#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_efa.py tests/test_golden.py -m gpu -x -q > gpurun_out/r02_t5.log 2>&1; tail -6 gpurun_out/r02_t5.log
timeout -k 10 300 python bench.py --no-cpu-baseline --hmc device > gpurun_out/r02_bench_efa.json 2> gpurun_out/r02_bench_efa.err; echo bench_efa rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --hmc host > gpurun_out/r02_bench_host.json 2> gpurun_out/r02_bench_host.err; echo bench_host rc=$?
timeout -k 10 200 python bench.py --no-cpu-baseline --hmc device --walkers-per-gpu 16 --streams 2 --matvec-reps 50 > gpurun_out/r02_bench_16x2.json 2>/dev/null; echo rc=$?
timeout -k 10 200 python bench.py --no-cpu-baseline --hmc device --walkers-per-gpu 16 --streams 1 --matvec-reps 50 > gpurun_out/r02_bench_16x1.json 2>/dev/null; echo rc=$?
echo done
