#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --timed-only > gpurun_out/dbg_a.json 2> gpurun_out/dbg_a.err; echo plain rc=$?
timeout -k 10 200 python bench.py --steps 4 --warmup 1 --timed-only > gpurun_out/dbg_b.json 2> gpurun_out/dbg_b.err; echo plain4 rc=$?
bash tools/profile_bench.sh r02; echo prof rc=$?; grep -c SIGSEGV gpurun_out/prof_r02.log
