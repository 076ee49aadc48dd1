"""fdm_wave_kernel (kernels_fdm_wave.hip, round 4): the fused MᵀM with one wavefront per run of slices and the whole time slice in
registers — rings (2 colours), plaquette lattices (4 colours), honeycomb lattices in 2 x 2 cell blocks (3 colours) — against the oracle's
mul_MtM! (src/FermionDetMatrix.jl:329-340), against the workgroup kernels it replaces (smoqy_matvec_wave(ctx, 0)), and through the CG
(twiddled operator, p·Ap partial = |Mp|²).  The host's lane-program detection is checked through smoqy_describe: lattices it must
accept and lattices it must refuse (those keep the workgroup kernels and must still be right)."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def models(kind, Ls, Lt, nw, hop):
    """hop: 'ssh' τ-dependent hoppings, 'uniform' t = 1 everywhere, 'bonds' τ-independent but different from bond to bond"""
    if kind == "honeycomb":
        ms = [lat.holstein_honeycomb(Ls, Lt, walker=w) for w in range(nw)]
    else:
        mk = lat.bssh_chain if kind == "chain" else lat.ossh_square
        ms = [mk(Ls, Lt, alpha=0.2 if hop == "ssh" else 0.0, walker=w) for w in range(nw)]
    if hop == "bonds":
        for w, m in enumerate(ms):
            tb = 1.0 + 0.2 * np.random.default_rng(100 + w).standard_normal(m.fpi.t.shape[0])
            m.fpi.t[...] = tb[:, None]
    return ms


def handle(ms):
    nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
    h = L.Handle(ms[0].fpi.Ltau, ms[0].fpi.N, nt, colors, True, len(ms), 1)
    oracles = []
    for w, m in enumerate(ms):
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
        oracles.append(orc.OracleFDM(nt, expV, ch, sh, True))
    return h, oracles


CASES = [
    # kind, L, Ltau, hoppings, lane program expected
    ("chain", 8, 12, "ssh", "ring"), ("chain", 24, 9, "ssh", "ring"), ("chain", 100, 5, "ssh", "ring"), ("chain", 252, 6, "ssh", "ring"), ("chain", 256, 8, "ssh", "ring"),
    ("chain", 256, 7, "uniform", "ring"), ("chain", 24, 13, "bonds", "ring"), ("chain", 10, 8, "ssh", None), ("chain", 260, 4, "ssh", None),
    ("square", 4, 12, "ssh", "plaquette"), ("square", 6, 13, "ssh", "plaquette"), ("square", 12, 6, "ssh", "plaquette"), ("square", 16, 4, "ssh", "plaquette"),
    ("square", 6, 8, "uniform", "plaquette"), ("square", 8, 5, "bonds", "plaquette"), ("square", 5, 6, "ssh", None), ("square", 18, 4, "ssh", None),
    ("honeycomb", 4, 12, "uniform", "honeycomb"), ("honeycomb", 6, 13, "uniform", "honeycomb"), ("honeycomb", 8, 5, "uniform", "honeycomb"), ("honeycomb", 16, 4, "uniform", "honeycomb"),
    ("honeycomb", 4, 8, "bonds", None),      # eight sites per lane: only hoppings that are uniform per colour take the wave kernel
    ("honeycomb", 3, 6, "uniform", None), ("honeycomb", 18, 4, "uniform", None),
]


@pytest.mark.parametrize("kind,Ls,Lt,hop,expect", CASES)
def test_wave_mtm_against_oracle_and_workgroup_kernels(kind, Ls, Lt, hop, expect):
    nw = 3
    ms = models(kind, Ls, Lt, nw, hop)
    h, oracles = handle(ms)
    N = ms[0].fpi.N
    g = np.random.default_rng(5)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    a, b, c = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    want = np.stack([oracles[w].mul_MtM(v[:, :, w]) for w in range(nw)], axis=2)
    for Tc in (1, 2):
        h.call("smoqy_set_tau_chunk", Tc)
        h.call("smoqy_matvec_wave", 0)
        h.call("smoqy_matvec_v", L.OP_MTM, b, a)
        ref = h.vec_download(b)
        assert "fdm_wave" not in h.describe()["mtm"]
        assert relerr(ref, want) < 1e-13
        for R in (-1, 1, 2, 3, 6, 64):   # automatic; runs that do not divide Lτ; a run longer than Lτ (rounded down to a multiple of the τ-chunk)
            h.call("smoqy_matvec_wave", R)
            h.call("smoqy_matvec_v", L.OP_MTM, c, a)
            got = h.vec_download(c)
            name = h.describe()["mtm"]
            took = "fdm_wave" in name
            if R == 1 and Tc == 2:
                assert not took              # a run shorter than the τ-chunk cannot keep the chunk layout of the p·Ap partials
            elif expect is None:
                assert not took, name
            elif R == -1:
                assert took == (expect == "plaquette"), name   # the automatic choice at three systems: plaquette lattices only (api_operator.hip, wave_run_length)
            else:
                assert took and expect in name, (name, R, Tc)
            assert relerr(got, want) < 1e-13, (kind, Ls, R, Tc, name)
            assert relerr(got, ref) < 1e-13
    # in place (out == in) keeps the workgroup kernels
    h.call("smoqy_matvec_wave", -1)
    h.call("smoqy_vec_copy", b, a)
    h.call("smoqy_matvec_v", L.OP_MTM, b, b)
    assert relerr(h.vec_download(b), want) < 1e-13
    # through the CG: twiddled operator (uniform hop phase, periodic τ), p·Ap from the kernel's |Mp|² partials
    h.call("smoqy_set_tau_chunk", 0)
    rv = np.ascontiguousarray(g.standard_normal((nw, N)))
    h.call("smoqy_precond_update_all", L.ptr(rv))
    its, sols = {}, {}
    for r in (0, 2):
        h.call("smoqy_matvec_wave", r)
        h.vec_upload(b, v)
        it, eps = np.zeros(nw, dtype=np.int32), np.zeros(nw)
        h.call("smoqy_cg_solve_v", b, b, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
        its[r], sols[r] = it.copy(), h.vec_download(b)
        assert np.all(eps < 1e-10)
    assert np.all(np.abs(its[0] - its[2]) <= 1), (its[0], its[2])
    assert relerr(sols[2], sols[0]) < 1e-8
    for w in range(nw):   # and it IS the solution: (MᵀM) x = b by the oracle
        assert relerr(oracles[w].mul_MtM(sols[2][:, :, w]), v[:, :, w]) < 1e-8
    h.close()


@pytest.mark.parametrize("name,nw", [("holstein_honeycomb_L16_Ltau128", 16), ("holstein_honeycomb_L16_Ltau128", 3), ("holstein_honeycomb_L8_Ltau80", 16), ("ossh_square_L12_Ltau100", 16),
                                     ("bssh_chain_L256_Ltau200", 16), ("bssh_chain_L256_Ltau200_alpha0p2", 5), ("holstein_honeycomb_L16_Ltau128", 64)])
def test_wave_mtm_at_the_benchmarked_shapes(name, nw):
    """the handles bench.py builds (fields formed on the device from the phonon fields): the wave kernel is what their CG launches"""
    batch = WalkerBatch(name, nwalkers=nw)
    h = batch.h
    g = np.random.default_rng(6)
    v = np.asfortranarray(g.standard_normal((batch.Lt, batch.N, nw)) + 1j * g.standard_normal((batch.Lt, batch.N, nw)))
    a, b, c = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    h.call("smoqy_matvec_wave", 0)
    h.call("smoqy_matvec_v", L.OP_MTM, b, a)
    assert "fdm_wave" not in h.describe()["mtm"]
    h.call("smoqy_matvec_wave", 2)   # forced: the automatic choice takes it for plaquette lattices and for 64 or more honeycomb systems only
    h.call("smoqy_matvec_v", L.OP_MTM, c, a)
    assert "fdm_wave" in h.describe()["mtm"], h.describe()
    ref, got = h.vec_download(b), h.vec_download(c)
    assert relerr(got, ref) < 1e-13
    for w in sorted({0, nw // 2, nw - 1}):
        m = batch.models[w]
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, batch.perm, m.fpi.dtau, True)
        o = orc.OracleFDM(batch.nt, expV, ch, sh, True)
        assert relerr(got[:, :, w], o.mul_MtM(v[:, :, w])) < 1e-13
    h.close()


_TWIN_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import test_gpu_wave_mtm as T
L = T.L
for Ls, Lt, nw in ((16, 8, 3), (4, 12, 3), (16, 6, 5)):
    ms = T.models("honeycomb", Ls, Lt, nw, "uniform")
    h, oracles = T.handle(ms)
    N = ms[0].fpi.N
    g = np.random.default_rng(7)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    want = np.stack([oracles[w].mul_MtM(v[:, :, w]) for w in range(nw)], axis=2)
    h.call("smoqy_set_tau_chunk", 2)
    for R in (2, 4):
        h.call("smoqy_matvec_wave", R)
        h.call("smoqy_matvec_v", L.OP_MTM, b, a)
        name = h.describe()["mtm"]
        assert "two wavefronts per SIMD" in name, name
        err = T.relerr(h.vec_download(b), want)
        assert err < 1e-13, (Ls, Lt, R, err)
    h.close()
print("TWIN_OK")
"""


def test_two_wave_twin_of_the_block_program_in_a_child_process():
    """SMOQY_FDM_WAVE_OCC=2 (read once per process, hence the child): the honeycomb-block program compiled for two wavefronts per SIMD
    (252 registers: exp(-ΔτV) held instead of the folded centre coefficients, which each propagate forms again; tests/test_kernel_resources.py
    holds the static counts) computes the same
    mul_MtM! (src/FermionDetMatrix.jl:329-340) — against the oracle at 1e-13 like every operator kernel.  Correctness only: the twin has not
    been timed yet and is off by default."""
    import os
    import subprocess
    import sys

    env = dict(os.environ, SMOQY_FDM_WAVE_OCC="2")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", _TWIN_CHILD, here], env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TWIN_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
