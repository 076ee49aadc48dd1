#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# a longer timed region: sweeps/s over 80 steps against the default 10 (drift, leaks: host RSS and device memory before / after)
python - <<'PY'
import json, os, subprocess, sys, time
def run(steps):
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-proc-scan", "--timed-only", "--steps", str(steps), "--warmup", "3"], capture_output=True, text=True)
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(f"steps {steps}: {d['value']:.1f} sweeps/s, {d['ms_per_step']:.1f} ms per step, avg iters {d['config']['avg_cg_iters']:.2f}, wall {time.time() - t0:.0f} s", flush=True)
run(10); run(80); run(10)
PY
