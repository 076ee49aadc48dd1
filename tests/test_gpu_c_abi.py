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
