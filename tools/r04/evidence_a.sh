#!/bin/bash
# round-4 evidence at HEAD, part A: full GPU suite, default bench, rocprofv3 profiles of the bench (timed region / roofline leg), solo profiles, gap probes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
t0=$SECONDS; timeout -k 10 600 python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
bash tools/profile_bench.sh r04 && echo prof ok
bash tools/profile_roofline.sh r04 && echo roof ok
bash tools/solo_profile.sh r04_hc16 16 && bash tools/solo_profile.sh r04_hc16_w1 1 && bash tools/solo_profile.sh r04_ossh 16 ossh_square_L12_Ltau100 && bash tools/solo_profile.sh r04_bssh 16 bssh_chain_L256_Ltau200 && bash tools/solo_profile.sh r04_hc8 16 holstein_honeycomb_L8_Ltau80 && echo solo ok
SMOQY_CHEB_WAVE=0 SMOQY_FDM_WAVE=0 bash tools/solo_profile.sh r04_ossh_wave_off 16 ossh_square_L12_Ltau100 && SMOQY_CHEB_WAVE=0 bash tools/solo_profile.sh r04_bssh_wave_off 16 bssh_chain_L256_Ltau200 && echo twins ok
SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r04_1walker 1 && SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r04_16walkers 16 && SMOQY_EFA=1 SMOQY_PREFETCH=1 SMOQY_ASYNC=0 bash tools/gap_probe.sh r04_1walker_polling 1 && echo gap ok
hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe > gpurun_out/r04_dpp_probe.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_output.json').read().strip().splitlines()[-1])
print('sweeps/s', d['value'], 'iters', d['avg_cg_iters'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['avg_cg_iters'], 'single', d['single_walker']['sweeps_per_s'], [round(x['sweeps_per_s'],1) for x in d['one_stream']])
PY
