#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python __graft_entry__.py smoke; echo smoke rc=$?
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-one-gpu --walkers-per-gpu 32 --streams 2 > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err; echo rehearse rc=$?
tail -c 600 gpurun_out/rehearse.json
python tools/nccl_probe.py; echo nccl rc=$?
