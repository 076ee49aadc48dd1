#!/bin/bash
# round 4: folded centre stage in the Chebyshev kernels — tests, regenerated pins (kept beside the old ones for the diff), solo profiles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_cheb_wave.py tests/test_gpu_parity.py tests/test_gpu_bench_shape.py tests/test_gpu_sweep_parity.py -m gpu -x -q > gpurun_out/r04_wave2_tests.log 2>&1; echo tests rc=$?; tail -5 gpurun_out/r04_wave2_tests.log
timeout -k 10 600 python tests/golden/device_cases.py > gpurun_out/device_cg_iterations.new.json 2> gpurun_out/device_cases.err; echo pins rc=$?
python - <<'PY'
import json
old=json.load(open('tests/golden/device_cg_iterations.json')); new=json.load(open('gpurun_out/device_cg_iterations.new.json'))
for k in sorted(new['device']):
    a,b=old['device'].get(k),new['device'][k]
    if a!=b:
        d=[y-x for x,y in zip(a,b)] if a else None
        print('moved', k, 'max |delta|', max(abs(v) for v in d) if d else None, 'oracle', new['oracle'].get(k))
PY
for spec in "bssh 16 bssh_chain_L256_Ltau200" "ossh 16 ossh_square_L12_Ltau100" "hc16 16 holstein_honeycomb_L16_Ltau128" "hc16_w1 1 holstein_honeycomb_L16_Ltau128" "hc8 16 holstein_honeycomb_L8_Ltau80"; do
  set -- $spec; bash tools/solo_profile.sh r04_$1 $2 $3 || exit 1; head -8 gpurun_out/solo_r04_$1.txt | cut -c1-150
done
