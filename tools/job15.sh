#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/trace_bench.sh r02 --timed-only --steps 2; cat gpurun_out/trace_r02.txt
