"""The plain-C caller examples/c_abi_demo.c (the ccall sequence of INTEGRATION.md in C) built with gcc and run on the GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_runs(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    lib_dir = os.path.join(ROOT, "smoqyelphqmc.jl_amd", "csrc")
    exe = tmp_path / "c_abi_demo"
    subprocess.run([gcc, "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L" + lib_dir, "-lsmoqy_hip", "-lm", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "adjoint identity" in r.stdout and "CG:" in r.stdout


def test_python_tutorial_example_runs():
    """examples/holstein_honeycomb_demo.py — the reference tutorial's call sequence against the mirror."""
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "holstein_honeycomb_demo.py"), "4", "20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "exactly 1" in r.stdout and "step 3" in r.stdout
    # G(r=0, τ=0) + G(r=0, τ=β) = 1 by construction of measure_GΔ0! (:221-227)
    tail = r.stdout.strip().splitlines()[-1]
    total = float(tail.split("sum = ")[1].split(",")[0])
    assert abs(total - 1.0) < 1e-12


def test_ranks_in_plain_c_join_a_served_team():
    """examples/walker_team_ranks.py: this process serves a walker team, K separately built plain-C programs (examples/team_member_demo.c,
    no GPU access) each drive one walker through smoqy_member_*: every rank reports its solves, and all ranks of a round see the same
    number of CG iterations only if their walkers happen to agree — what must agree is the count of solves, 2 + Nt + 1 per sweep."""
    import json
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "walker_team_ranks.py"), "holstein_honeycomb_L4_Ltau40", "3", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert sorted(q["walker"] for q in recs) == [0, 1, 2]
    for q in recs:
        assert q["solves"] == 2 * (2 + 24 + 1) and q["cg_iterations"] > q["solves"] and abs(q["last_dH"]) < 50.0
    assert "sweeps/s on one GPU" in r.stdout
