#!/bin/bash
# round-3 evidence, part B: PMC passes (counters only, --kernel-trace for the names)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for nb in 16 64 128; do bash tools/pmc_traffic.sh $nb > gpurun_out/r03_pmc_traffic_$nb.txt 2>&1; cp gpurun_out/pmc_traffic_fdm_mtm_b$nb.json gpurun_out/r03_pmc_traffic_fdm_mtm_b$nb.json; tail -4 gpurun_out/r03_pmc_traffic_$nb.txt; done
bash tools/pmc_iteration.sh > gpurun_out/r03_pmc_iteration.txt 2>&1; cp gpurun_out/pmc_iteration.json gpurun_out/r03_pmc_iteration.json; tail -6 gpurun_out/r03_pmc_iteration.txt
bash tools/pmc_sq.sh "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" lds_r03 > gpurun_out/r03_pmc_lds_iteration.txt 2>&1; tail -30 gpurun_out/r03_pmc_lds_iteration.txt
SMOQY_CHEB_WL0=0 bash tools/pmc_sq.sh "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" lds_r03_wl0off > gpurun_out/r03_pmc_lds_iteration_wl0off.txt 2>&1
PMC_SET=sq bash tools/pmc_explore.sh r03_sq_mtm_b128 fdm_ -- tools/matvec_only.py 128 20 > /dev/null 2>&1; cp gpurun_out/pmc_explore_r03_sq_mtm_b128.txt gpurun_out/r03_pmc_explore_sq_mtm_b128.txt; head -30 gpurun_out/r03_pmc_explore_sq_mtm_b128.txt
echo done
