#!/bin/bash
# round 4: the asynchronous trajectory only where it pays (short solves, back-off after a miss): EFA + team tests, then the other lattices' bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_efa.py tests/test_gpu_team.py -m gpu -x -q > gpurun_out/r04_async_policy_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 gpurun_out/r04_async_policy_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/r04/configs_bench.sh
python - <<'PY'
import json
for wl in ["holstein_honeycomb_L8_Ltau80","ossh_square_L12_Ltau100","bssh_chain_L256_Ltau200","holstein_honeycomb_L4_Ltau40"]:
    d=json.loads(open(f'gpurun_out/r04_bench_{wl}.json').read().strip().splitlines()[-1]); print(wl, d['config']['hmc_async'])
PY
