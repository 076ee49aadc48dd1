#!/bin/bash
# round-4 evidence at HEAD, part B: PMC passes (counters only + --kernel-trace, a few counters per pass)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for nb in 16 64 128; do bash tools/pmc_traffic.sh $nb > /dev/null && echo traffic b$nb ok; done
PMC_TAG=hc16 bash tools/pmc_iteration.sh | cut -c1-150
L="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
bash tools/pmc_sq.sh "$L" r04_lds_hc16 > /dev/null && echo sq hc16 ok
PMC_WORKLOAD=bssh_chain_L256_Ltau200 bash tools/pmc_sq.sh "$L" r04_lds_bssh > /dev/null && PMC_WORKLOAD=ossh_square_L12_Ltau100 bash tools/pmc_sq.sh "$L" r04_lds_ossh > /dev/null && echo sq ssh ok
SMOQY_CHEB_WAVE=0 PMC_WORKLOAD=bssh_chain_L256_Ltau200 bash tools/pmc_sq.sh "$L" r04_lds_bssh_wave_off > /dev/null && SMOQY_CHEB_WAVE=0 SMOQY_FDM_WAVE=0 PMC_WORKLOAD=ossh_square_L12_Ltau100 bash tools/pmc_sq.sh "$L" r04_lds_ossh_wave_off > /dev/null && echo sq twins ok
PMC_SET=sq bash tools/pmc_explore.sh r04_sq_mtm_b128 fdm_ -- tools/matvec_only.py 128 20 > /dev/null; echo explore rc=$?
SMOQY_FDM_WAVE_R=16 PMC_SET=sq bash tools/pmc_explore.sh r04_sq_mtm_b128_wave fdm_ -- tools/matvec_only.py 128 20 > /dev/null; echo explore wave rc=$?
ls gpurun_out | grep -E "pmc_(sq|explore|traffic|iteration)"
