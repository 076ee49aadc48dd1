"""Both kernel families stay covered: the owner-computes kernels are the default for Sym, so this re-runs the operator / KPM / CG
parity tests in a child process with them switched off (SMOQY_CHEB_OWN=0, SMOQY_FDM_OWN=0: the LDS-resident twins)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suite_with_the_lds_resident_kernels():
    env = dict(os.environ, SMOQY_CHEB_OWN="0", SMOQY_FDM_OWN="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_irregular.py"), "-m", "gpu", "-x", "-q"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_parity_suite_with_a_chebyshev_workgroup_per_frequency():
    """SMOQY_CHEB_GROUP=1: every frequency gets its own Chebyshev workgroup(s) (the form before the light workgroups of cheb_own_kernel,
    kept as the A/B twin) — same parity files, same tolerances."""
    env = dict(os.environ, SMOQY_CHEB_GROUP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
