#!/bin/bash
# round 4: the default bench (its per-workload best shape, CPU baseline included) on the other BASELINE.json lattices, after the PMC passes
# of the fused MtM at the systems-per-launch those shapes use -> gpurun_out/r04_bench_<workload>.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/pmc_traffic.sh 64 holstein_honeycomb_L8_Ltau80 hc8 > /dev/null && echo pmc b64 hc8 ok
bash tools/pmc_traffic.sh 64 ossh_square_L12_Ltau100 ossh > /dev/null && echo pmc b64 ossh ok
mkdir -p profiles_new; cp gpurun_out/pmc_traffic_fdm_mtm_b*_*.json gpurun_out/pmc_iteration_*.json profiles_new/ 2>/dev/null
# bench.py reads the committed PMC files from profiles/: give this run the ones just taken
for f in gpurun_out/pmc_traffic_fdm_mtm_b*_*.json; do cp $f profiles/r04_$(basename $f); done
for f in gpurun_out/pmc_iteration_*.json; do [ -e "$f" ] && cp $f profiles/r04_$(basename $f); done
for wl in ${WORKLOADS:-holstein_honeycomb_L8_Ltau80 ossh_square_L12_Ltau100 bssh_chain_L256_Ltau200 holstein_honeycomb_L4_Ltau40}; do
  t0=$SECONDS
  timeout -k 10 500 python bench.py --workload $wl --steps ${STEPS:-6} --warmup 2 --no-proc-scan > gpurun_out/r04_bench_$wl.json 2> gpurun_out/r04_bench_$wl.err; echo $wl rc=$? wall $((SECONDS-t0)) s
  python - $wl <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r04_bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['cpu_baseline']
print(sys.argv[1], 'sweeps/s', round(d['value'],1), 'shape', d['config']['walkers_per_gpu'], 'x', d['config']['streams_per_gpu'], 'iters', round(d['avg_cg_iters'],1), 'cpu', round(c['value'],2), round(c['avg_cg_iters'],1), 'one_stream', [round(x['sweeps_per_s'],1) for x in d['one_stream']])
print('   roofline', r['kernel'][:40], 'frac', round(r['frac'],3), 'single_pass', round(r['frac_single_pass'],3), 'traffic', r['frac_traffic'] and round(r['frac_traffic'],3), 'us', round(r['avg_launch_us'],1), 'iso', round(r['isolated']['avg_launch_us'],1), 'cgtraffic', d.get('cg_iteration_traffic',{}).get('frac'))
PY
done
