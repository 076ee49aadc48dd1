#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for g in 0 1; do
 v=$(SMOQY_BENCH_GRAPH=$g timeout -k 10 300 python bench.py --timed-only --steps 6 --no-mtm-sampling 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
 echo "graph=$g: $v"
done; done
