#!/bin/bash
# round-4 evidence at the final HEAD (tfft_rb_kernel in): full GPU suite, default bench, rocprofv3 stats of the timed region, chain bench record,
# PMC passes of the iteration on the three lattices whose tau-FFT changed (traffic; LDS counters for the chain and the square lattice)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r04_gpu_suite.log 2>&1; echo suite rc=$?; tail -3 gpurun_out/r04_gpu_suite.log
t0=$SECONDS; timeout -k 10 600 python bench.py > gpurun_out/r04_bench_output.json 2> gpurun_out/r04_bench_output.err; echo bench rc=$? wall $((SECONDS-t0)) s
bash tools/profile_bench.sh r04 && echo prof ok
PMC_TAG=bssh PMC_WORKLOAD=bssh_chain_L256_Ltau200 bash tools/pmc_iteration.sh | cut -c1-150
PMC_TAG=ossh PMC_WORKLOAD=ossh_square_L12_Ltau100 bash tools/pmc_iteration.sh | cut -c1-150
PMC_TAG=hc8 PMC_WORKLOAD=holstein_honeycomb_L8_Ltau80 bash tools/pmc_iteration.sh | cut -c1-150
L="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
PMC_WORKLOAD=bssh_chain_L256_Ltau200 bash tools/pmc_sq.sh "$L" r04_lds_bssh > /dev/null && PMC_WORKLOAD=ossh_square_L12_Ltau100 bash tools/pmc_sq.sh "$L" r04_lds_ossh > /dev/null && echo sq ok
for f in gpurun_out/pmc_iteration_bssh.json gpurun_out/pmc_iteration_ossh.json gpurun_out/pmc_iteration_hc8.json; do cp $f profiles/r04_$(basename $f); done
for wl in bssh_chain_L256_Ltau200 ossh_square_L12_Ltau100 holstein_honeycomb_L8_Ltau80; do
  t0=$SECONDS
  timeout -k 10 500 python bench.py --workload $wl --steps 6 --warmup 2 --no-proc-scan > gpurun_out/r04_bench_$wl.json 2> gpurun_out/r04_bench_$wl.err; echo $wl rc=$? wall $((SECONDS-t0)) s
done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_output.json').read().strip().splitlines()[-1])
print('sweeps/s', d['value'], 'iters', d['avg_cg_iters'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['avg_cg_iters'], 'single', d['single_walker']['sweeps_per_s'], [round(x['sweeps_per_s'],1) for x in d['one_stream']], 'roofline', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1))
for wl in ['bssh_chain_L256_Ltau200','ossh_square_L12_Ltau100','holstein_honeycomb_L8_Ltau80']:
    d=json.loads(open(f'gpurun_out/r04_bench_{wl}.json').read().strip().splitlines()[-1])
    print(wl, round(d['value'],1), d['config'].get('tfft_kernel'), [round(x['sweeps_per_s'],1) for x in d['one_stream']], 'traffic frac', d.get('cg_iteration_traffic',{}).get('frac'))
PY
