#!/bin/bash
# default bench on the other BASELINE.json lattices (device EFA trajectory, 128 walkers on 8 streams) -> gpurun_out/r03_bench_<workload>.json
cd $GRAFT_REPO_ROOT
for wl in ${WORKLOADS:-holstein_honeycomb_L4_Ltau40 holstein_honeycomb_L8_Ltau80 ossh_square_L12_Ltau100 bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100_alpha0p2 bssh_chain_L256_Ltau200_alpha0p2}; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-proc-scan --workload $wl --steps ${STEPS:-6} --warmup ${WARMUP:-2} > gpurun_out/r03_bench_$wl.json 2> gpurun_out/r03_bench_$wl.err; echo $wl rc=$?
  python - $wl <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r03_bench_{sys.argv[1]}.json')); print(sys.argv[1], round(d['value'],1), round(d['config']['avg_cg_iters'],1), [round(x['sweeps_per_s'],1) for x in d['one_stream']], round(d['roofline']['frac'],2), round(d['roofline']['avg_launch_us'],1))
PY
done
