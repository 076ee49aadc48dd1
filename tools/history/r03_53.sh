#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/collect_evidence_r03b.sh > gpurun_out/r03_53_pmc.out 2>&1; tail -3 gpurun_out/r03_53_pmc.out
bash tools/config_scan.sh "128 8 1 4" "128 8 1 3" "128 8 1 5" "192 8 1 4" "256 8 1 4" "128 4 1 4" "192 6 1 4" > gpurun_out/r03_config_scan.txt 2>&1; cat gpurun_out/r03_config_scan.txt
