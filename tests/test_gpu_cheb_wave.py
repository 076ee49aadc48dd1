"""cheb_wave_kernel (kernels_kpm_wave.hip, round 4): the Sym Chebyshev apply with one wavefront per chain — rings (2 colours) and
plaquette lattices (4 colours) of up to 256 sites — against the oracle's ldiv!(u', P, u) (src/KPMPreconditioner.jl:355-414), through the
preconditioned CG, and against the owner-computes kernel it replaces (SMOQY_CHEB_WAVE=0, child process).  The host's lane-program
detection (api_handle.hip, wave_program) is checked through smoqy_traits: lattices it must accept, lattices it must refuse."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # kind, L, Ltau, (wave_kind, lanes) the host must find
    ("chain", 8, 24, (1, 2)),
    ("chain", 100, 40, (1, 25)),
    ("chain", 252, 16, (1, 63)),      # 63 lanes: the rotation by one lane goes through ds_bpermute
    ("chain", 256, 40, (1, 64)),      # BASELINE config 5's lattice: DPP wave rotations
    ("chain", 10, 12, (0, 0)),        # N = 10 is not a multiple of four
    ("chain", 260, 8, (0, 0)),        # more than 64 lanes' worth of sites
    ("chain", 9, 12, (0, 0)),         # odd ring: three colours
    ("square", 4, 24, (2, 4)),
    ("square", 8, 16, (2, 16)),
    ("square", 12, 40, (2, 36)),      # BASELINE config 3's lattice
    ("square", 16, 12, (2, 64)),
    ("square", 5, 12, (0, 0)),        # odd L: more than four colours
    ("square", 18, 8, (0, 0)),        # 81 plaquettes
]


def build(kind, Ls, Lt, nw=2):
    mk = lat.bssh_chain if kind == "chain" else lat.ossh_square
    models = [mk(Ls, Lt, walker=w) for w in range(nw)]
    nt, perm, colors = lat.checkerboard_decomposition(models[0].fpi.neighbor_table)
    h = L.Handle(Lt, models[0].fpi.N, nt, colors, True, nw, 1)
    oracles = []
    for w, m in enumerate(models):
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
        h.call("smoqy_update_fields", w, L.ptr(expV), L.ptr(ch), L.ptr(sh))
        oracles.append(orc.OracleFDM(nt, expV, ch, sh, True))
    return h, oracles, models[0].fpi.N


@pytest.mark.parametrize("kind,Ls,Lt,expect", CASES)
def test_wave_program_detection_apply_and_cg(kind, Ls, Lt, expect):
    nw = 2
    h, oracles, N = build(kind, Ls, Lt, nw)
    tr = h.traits()
    assert (tr["wave_kind"], tr["wave_lanes"]) == expect, tr
    g = np.random.default_rng(11)
    Ps = []
    for w in range(nw):
        rv = g.standard_normal(N)
        P = orc.OracleKPM(oracles[w])
        P.update(rv)
        h.call("smoqy_precond_update", w, L.ptr(rv))
        assert P.active
        Ps.append(P)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, nw)
    for s in range(nw):
        want = Ps[s].apply(v[:, :, s])
        assert np.abs(out[:, :, s] - want).max() < 1e-11 * np.abs(want).max(), (kind, Ls, s)
    name = h.describe()["cheb"]
    assert ("cheb_wave_kernel" in name) == (expect[0] != 0), name
    # through the CG: same iteration count (one step of slack for the other summation order of r·z), same solution
    x = np.zeros_like(v)
    it = np.zeros(nw, dtype=np.int32)
    eps = np.zeros(nw)
    h.call("smoqy_cg_solve", L.ptr(x), L.ptr(v), 1, 0, nw, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
    for s in range(nw):
        xo, ito, _ = oracles[s].cg_solve(v[:, :, s], precond=Ps[s], tol=1e-10, maxiter=10000)
        assert abs(int(it[s]) - ito) <= 1, (it, ito)
        assert np.abs(x[:, :, s] - xo).max() < 1e-8 * np.abs(xo).max()
    h.close()


def test_real_vector_apply_keeps_the_workgroup_kernel():
    """The real-vector ldiv! (half the frequencies, no component split) is not a wave-kernel case: it must keep cheb_own_kernel and its
    parity (tests/test_gpu_parity.py covers the values; here: the dispatch)."""
    h, oracles, N = build("chain", 256, 16, 1)
    rv = np.random.default_rng(3).standard_normal(N)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    P = orc.OracleKPM(oracles[0])
    P.update(rv)
    u = np.asfortranarray(np.random.default_rng(4).standard_normal((16, N, 1)))
    out = np.zeros_like(u)
    h.call("smoqy_precond_apply_real", L.ptr(out), L.ptr(u), 0, 1)
    want = P.apply_real(u[:, :, 0])
    assert np.abs(out[:, :, 0] - want).max() < 1e-11 * np.abs(want).max()
    h.close()


@pytest.mark.parametrize("mode", ["0", "2"])  # 0: cheb_own_kernel everywhere; 2: rings of 64 lanes through ds_bpermute instead of the DPP wave rotations
def test_the_twins_stay_covered(mode):
    env = dict(os.environ, SMOQY_CHEB_WAVE=mode)
    if mode == "0":
        args = [os.path.join(ROOT, "tests", "test_gpu_bench_shape.py"), "-k", "(ossh or bssh) and not switched_off"]
    else:
        args = [os.path.join(ROOT, "tests", "test_gpu_cheb_wave.py"), "-k", "detection and chain-256"]
    r = subprocess.run([sys.executable, "-m", "pytest", *args, "-m", "gpu", "-x", "-q"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
