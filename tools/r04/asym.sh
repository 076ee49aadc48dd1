#!/bin/bash
# round 4: where the Asym forms (AsymFermionDetMatrix / AsymKPMPreconditioner) stand against Sym: one-stream sweeps and solo kernel durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for nw in 1 16; do for form in sym asym; do echo "$nw walkers $form: $(SMOQY_EFA=1 SMOQY_PREFETCH=1 python tools/one_stream.py $nw holstein_honeycomb_L16_Ltau128 $form | tail -1)"; done; done | tee gpurun_out/r04_asym_scan.txt
bash tools/solo_profile.sh r04_hc16_asym 16 holstein_honeycomb_L16_Ltau128 asym && head -12 gpurun_out/solo_r04_hc16_asym.txt | cut -c1-170
