#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for r in 0 2 4; do
 v=$(SMOQY_FDM_WAVE_R=$r timeout -k 10 300 python bench.py --timed-only --steps 6 --no-mtm-sampling 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
 echo "wave R=$r: $v"
done; done
