#!/bin/bash
# (historical: removed again, docs/DESIGN_LOG.md R4.14) cheb_wave_kernel<plaquette, RELOAD> (125 VGPRs, four wavefronts per SIMD, no scratch: SMOQY_CHEB_WAVE=3 of that build) against the 153-VGPR form: optical-SSH bench at 4 x 64 and one stream at 16 / 64
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 env SMOQY_CHEB_WAVE=3 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "ossh" > gpurun_out/r04_cheb_cap_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -3 gpurun_out/r04_cheb_cap_tests.log
[ $rc -ne 0 ] && exit $rc
for e in 1 3 1 3; do
  SMOQY_CHEB_WAVE=$e timeout -k 10 300 python bench.py --workload ossh_square_L12_Ltau100 --timed-only --steps 6 --warmup 2 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ossh_square_L12_Ltau100 4 x 64 SMOQY_CHEB_WAVE=$e', round(d['value'],1), 'sweeps/s')" || exit 1
done | tee gpurun_out/r04_cheb_reload_scan.txt
for nw in 16 64; do
  for e in 1 3 1 3; do
    echo "ossh one stream $nw walkers SMOQY_CHEB_WAVE=$e $(SMOQY_SWEEPS=2 SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_SPLIT=0 SMOQY_CHEB_WAVE=$e timeout -k 10 300 python tools/one_stream.py $nw ossh_square_L12_Ltau100 2> /dev/null | tail -1)"
  done
done | tee -a gpurun_out/r04_cheb_reload_scan.txt
