cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_suite3.log 2>&1; tail -4 gpurun_out/r02_gpu_suite3.log
for cfg in "1" "2"; do
  timeout -k 10 300 python bench.py --timed-only --steps 8 --cg-split $cfg --no-mtm-sampling > gpurun_out/split_$cfg.json 2>/dev/null
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/split_{sys.argv[1]}.json')); print('96x6 cg-split',sys.argv[1], round(d['value'],1))
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 > gpurun_out/split_os.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/split_os.json'));print('one_stream (auto split)', [round(x['sweeps_per_s'],1) for x in d['one_stream']], d['value'])"
