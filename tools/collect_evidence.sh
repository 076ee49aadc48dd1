#!/bin/bash
# final round-2 evidence at HEAD: full GPU suite, default bench, profiles
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_suite.log 2>&1; tail -4 gpurun_out/r02_gpu_suite.log
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_output.json 2> gpurun_out/r02_bench_output.err; echo bench rc=$?
bash tools/profile_bench.sh r02 && echo prof ok
bash tools/profile_roofline.sh r02 && echo roof ok
bash tools/solo_profile.sh r02_hc16 16 && bash tools/solo_profile.sh r02_ossh 16 ossh_square_L12_Ltau100 && bash tools/solo_profile.sh r02_bssh 16 bssh_chain_L256_Ltau200 && bash tools/solo_profile.sh r02_hc8 16 holstein_honeycomb_L8_Ltau80
bash tools/solo_profile.sh r02_hc16_asym 16 holstein_honeycomb_L16_Ltau128 asym && bash tools/solo_profile.sh r02_hc16_w1 1
bash tools/pmc_iteration.sh > gpurun_out/r02_pmc_iteration.txt 2>&1; cp gpurun_out/pmc_iteration.json gpurun_out/r02_pmc_iteration.json
bash tools/pmc_sq.sh "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" lds_r02 > gpurun_out/r02_pmc_lds_iteration.txt 2>&1
echo done
