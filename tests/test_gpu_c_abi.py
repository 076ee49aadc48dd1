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
