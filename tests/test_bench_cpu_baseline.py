"""bench.py's CPU-baseline leg (no GPU involved): the oracle sample a worker process times, and the one-walker-per-core
aggregation, on the smallest BASELINE.json lattice."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_worker_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", "1", "--workload", "holstein_honeycomb_L4_Ltau40"], capture_output=True, text=True, timeout=300, check=True)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["value"] > 0 and d["cores"] == 1 and d["kind"] == "port" and "walker 1" in d["sample"]


def test_cpu_baseline_aggregates_over_the_available_cores():
    sys.path.insert(0, ROOT)
    import bench

    n = bench.available_cores()
    assert 1 <= n <= 64
    r = bench.cpu_baseline("holstein_honeycomb_L4_Ltau40", 1e-10, 24)
    assert r["cores"] == n and r["value"] > 0
    assert r["value"] <= n * r["per_core_max"] * (1 + 1e-12) and r["per_core_min"] <= r["per_core_max"]
    assert abs(r["value"] - r["single_core_value"]) >= 0 and r["unit"] == "sweeps/s"
