#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_team.py -m gpu -q -x > gpurun_out/r03_team_test.log 2>&1; echo "team test rc=$?"; tail -15 gpurun_out/r03_team_test.log
timeout -k 10 300 python - <<'PY'
import sys, json
sys.path.insert(0,'.')
import bench
class A: pass
a=A(); a.workload="holstein_honeycomb_L16_Ltau128"; a.scan_sweeps=4
print(json.dumps(bench.team_scan(a,[8,16,32],0,0)))
PY
