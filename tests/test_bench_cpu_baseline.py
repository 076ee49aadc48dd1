"""bench.py's CPU-baseline leg (no GPU involved): the oracle sample a worker process times, and the one-walker-per-core
aggregation, on the smallest BASELINE.json lattice."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_worker_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "1", "--workload", "holstein_honeycomb_L4_Ltau40"], capture_output=True, text=True, timeout=300, check=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["value"] > 0 and d["cores"] == 1 and d["kind"] == "port" and "walker 1" in d["sample"]
    # work-normalised fields: the sample is whole sweeps of 27 solves, iterations per second = avg iterations x 27 x sweeps per second
    assert d["solves_per_sweep"] == 27 and 5 < d["avg_cg_iters"] < 30
    assert abs(d["cg_iterations_per_s"] - d["avg_cg_iters"] * 27 * d["value"]) < 1e-9 * d["cg_iterations_per_s"]


def test_cpu_baseline_aggregates_over_the_available_cores():
    sys.path.insert(0, ROOT)
    import bench

    n = bench.available_cores()
    assert 1 <= n <= 64
    r = bench.cpu_baseline("holstein_honeycomb_L4_Ltau40", 1e-10, 24)
    assert r["cores"] == n and r["value"] > 0
    assert r["value"] <= n * r["per_core_max"] * (1 + 1e-12) and r["per_core_min"] <= r["per_core_max"]
    assert abs(r["value"] - r["single_core_value"]) >= 0 and r["unit"] == "sweeps/s"
    assert r["cg_iterations_per_s"] > 0 and 5 < r["avg_cg_iters"] < 30


@pytest.mark.gpu
def test_cpu_sample_and_gpu_sweep_are_the_same_work():
    """the CPU baseline follows the trajectory the GPU sweep follows.  Same walkers, same random streams: the
    average iterations per solve of the two legs agree within 2 % (they are equal solve by solve up to +-1, tests/test_gpu_sweep_parity.py)."""
    sys.path.insert(0, ROOT)
    import bench
    from smoqyelphqmc_amd.walkers import WalkerBatch

    name = "holstein_honeycomb_L4_Ltau40"
    cpu = [bench.cpu_sample(name, 1e-10, 24, walker=w) for w in range(2)]
    b = WalkerBatch(name, nwalkers=2, walker0=0, device_efa=True, prefetch_randoms=True)
    for _ in range(bench.CPU_SWEEPS):
        b.sweep()
    b.h.call("smoqy_sync")
    gpu_avg = b.stats.iters_sum / b.stats.solves
    cpu_avg = sum(c["avg_cg_iters"] for c in cpu) / 2
    b.h.close()
    assert abs(cpu_avg - gpu_avg) <= 0.02 * gpu_avg, (cpu_avg, gpu_avg)
