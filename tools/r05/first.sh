#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4, when no GPU minutes were left; docs/DESIGN_LOG.md R4.15).
# A/B of the honeycomb-block fdm_wave_kernel: default (264 registers, one wavefront per SIMD) against the twin of SMOQY_FDM_WAVE_OCC=2
# (256 registers, two wavefronts per SIMD), same box, interleaved, then the twin's correctness test and the bench with the wave kernel
# forced onto 16-system launches.  usage: tools/gpu.sh 900 'bash tools/r05/first.sh'
set -e -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05_first; mkdir -p $out
python -m pytest tests/test_gpu_wave_mtm.py -x -q -k "twin or honeycomb" > $out/tests.log 2>&1
for rep in 1 2; do
  for occ in 1 2; do
    SMOQY_FDM_WAVE_OCC=$occ timeout -k 10 240 python tools/wave_scan.py holstein_honeycomb_L16_Ltau128 16,32,64,128 0,-1,2,4,8,16,32 2 > $out/wave_scan_occ${occ}_rep${rep}.txt 2>&1
  done
done
# the eight-stream bench: default, and the wave kernel on its 16-system launches with one and with two wavefronts per SIMD
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
for occ in 1 2; do
  SMOQY_FDM_WAVE_OCC=$occ SMOQY_FDM_WAVE_R=4 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_wave_occ${occ}.json 2> $out/bench_wave_occ${occ}.err
done
tail -n 3 $out/wave_scan_occ*_rep2.txt
