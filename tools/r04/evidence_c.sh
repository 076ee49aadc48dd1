#!/bin/bash
# round-4 evidence at HEAD, part C: N > 1 control flow rehearsed on the one GPU (2 and 4 ranks share cuda:0; gloo barrier / MAX; no scaling claim)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --steps 3 --warmup 1 --walkers-per-gpu 64 --streams 4 > gpurun_out/r04_rehearse_2ranks.json 2> gpurun_out/r04_rehearse_2ranks.err; echo rehearse2 rc=$?; tail -c 300 gpurun_out/r04_rehearse_2ranks.json | cut -c1-300
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 4 --rehearse-one-gpu --steps 3 --warmup 1 --walkers-per-gpu 32 --streams 2 > gpurun_out/r04_rehearse_4ranks.json 2> gpurun_out/r04_rehearse_4ranks.err; echo rehearse4 rc=$?; tail -c 300 gpurun_out/r04_rehearse_4ranks.json | cut -c1-300
python - <<'PY'
import json
for n in (2,4):
    d=json.loads(open(f'gpurun_out/r04_rehearse_{n}ranks.json').read().strip().splitlines()[-1]); print(n, 'ranks', d['n_gpus'], round(d['value'],1), d['config']['parallelism'])
PY
python tools/nccl_probe.py 2>&1 | tail -2
