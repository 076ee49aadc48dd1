#!/bin/bash
cd $GRAFT_REPO_ROOT
for xm in 0 1; do SMOQY_XCD_MAP=$xm PMC_WALKERS=8 bash tools/pmc_iteration.sh > gpurun_out/pmci8_$xm.txt 2>&1; echo "xcd_map=$xm, 8 walkers:"; python3 -c "
import json
d=json.load(open('gpurun_out/pmc_iteration.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'traffic_MB' in v: print('  ', k[:70], round(v['traffic_MB'],1), 'fetch', round(v['FETCH_SIZE']['median_KB']/512,1))
"; done
for xm in 0 1; do SMOQY_XCD_MAP=$xm PMC_WALKERS=16 PMC_WORKLOAD=bssh_chain_L256_Ltau200_alpha0p2 bash tools/pmc_iteration.sh > gpurun_out/pmcib_$xm.txt 2>&1; echo "xcd_map=$xm, bssh 16 walkers:"; python3 -c "
import json
d=json.load(open('gpurun_out/pmc_iteration.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'traffic_MB' in v: print('  ', k[:70], round(v['traffic_MB'],1), 'fetch', round(v['FETCH_SIZE']['median_KB']/512,1))
"; done
