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


@pytest.mark.parametrize("wl0", ["0", "1"])  # 0: LDS exchange (round 2); 1: ds_bpermute (the wave-local form before the DPP row rotations)
def test_bench_shape_with_the_round_2_twins_of_the_round_3_kernels(wl0):
    """Round 3 added a streaming MᵀM kernel (from 16 systems per launch) and a wave-local exchange in the Chebyshev chain (lattices whose
    colour-0 mates sit inside a wavefront: the headline lattice).  Their round-2 forms — the chunked MᵀM kernel and the LDS exchange — stay
    selectable (SMOQY_FDM_STREAM=0, SMOQY_CHEB_WL0=0) and are re-run here at the benchmarked shape of the headline lattice against the
    oracle, same tolerances."""
    # SMOQY_TFFT_EDGE=0: the Stockham / in-place pass schedules instead of the register-blocked τ-FFT that Lτ = 128 now takes
    env = dict(os.environ, SMOQY_FDM_STREAM="0", SMOQY_CHEB_WL0=wl0, SMOQY_TFFT_EDGE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_bench_shape.py"), "-m", "gpu", "-x", "-q", "-k", "honeycomb and 16sys and not switched_off"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("mode", ["1", "0"])
def test_streaming_mtm_on_the_lane_program(mode):
    """fdm_own_stream_kernel (the streaming MᵀM with the owner-computes stage chain) is chosen for handles of 32 systems or more;
    SMOQY_FDM_OWNSTREAM=1 selects it wherever fdm_stream_kernel would run on a 256-lane lattice with τ-independent hoppings, =0 never: the
    streaming tests and the bench shapes against the oracle with each kernel at every size, same tolerances."""
    env = dict(os.environ, SMOQY_FDM_OWNSTREAM=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_stream_mtm.py"), os.path.join(ROOT, "tests", "test_gpu_bench_shape.py"),
                        "-m", "gpu", "-x", "-q", "-k", "not switched_off"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
