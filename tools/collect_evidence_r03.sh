#!/bin/bash
# round-3 evidence at HEAD, part A: full GPU suite, default bench, rocprofv3 profiles (what profiles/r03_* were produced by)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r03_gpu_suite.log 2>&1; tail -3 gpurun_out/r03_gpu_suite.log
t0=$SECONDS; timeout -k 10 500 python bench.py > gpurun_out/r03_bench_output.json 2> gpurun_out/r03_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
bash tools/profile_bench.sh r03 && echo prof ok
bash tools/profile_roofline.sh r03 && echo roof ok
bash tools/solo_profile.sh r03_hc16 16 && bash tools/solo_profile.sh r03_hc16_w1 1 && bash tools/solo_profile.sh r03_ossh 16 ossh_square_L12_Ltau100 && bash tools/solo_profile.sh r03_bssh 16 bssh_chain_L256_Ltau200 && bash tools/solo_profile.sh r03_hc8 16 holstein_honeycomb_L8_Ltau80
SMOQY_CHEB_WL0=0 bash tools/solo_profile.sh r03_hc16_wl0off 16 && SMOQY_CHEB_WL0=0 bash tools/solo_profile.sh r03_hc16_w1_wl0off 1
SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r03_1walker 1 && SMOQY_EFA=1 SMOQY_PREFETCH=1 bash tools/gap_probe.sh r03_16walkers 16
echo done
# N > 1 control flow rehearsed on the one GPU: two ranks share cuda:0, gloo for the barrier and the MAX (no scaling claim)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --walkers-per-gpu 64 --streams 4 > gpurun_out/r03_rehearse_2ranks.json 2> gpurun_out/r03_rehearse_2ranks.err; echo rehearse rc=$?; tail -c 600 gpurun_out/r03_rehearse_2ranks.json
# walker-team scans on their own (more points than the bench record holds)
timeout -k 10 500 python tools/team_procs_scan.py 16,32,64,2x32,4x32 4 > gpurun_out/r03_team_procs_devhmc.txt 2>&1; timeout -k 10 500 python tools/team_scan.py 16,32,64 4 2x32,4x32 > gpurun_out/r03_team_scan_devhmc.txt 2>&1; tail -2 gpurun_out/r03_team_procs_devhmc.txt | cut -c1-200
