"""The five BASELINE.json configurations at FULL size on the GPU, checked through
size-independent properties (and against the CPU oracle where it finishes in seconds):

  * adjoint identity  <u, M v> = <Mᵀ u, v>  and  MᵀM = Mᵀ(M ·)  across separate launches
  * KPM-preconditioned CG: residual recomputed with an independent MᵀM apply, same solution as the
    unpreconditioned solve, rtol 1e-10 (BASELINE.json north_star)
  * heat-bath identity of the pseudofermion action: Φ = Λᵀ Mᵀ R  ==>  S_f = |R|²
  * FourierTransformer round trip / unitarity at the config's Ltau (40, 80, 100, 128, 200)
"""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice

CONFIGS = list(lat.CONFIGS)


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("name", CONFIGS)
def test_full_size_operator_properties(name):
    m = lat.CONFIGS[name]()
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N = m.fpi.Ltau, m.fpi.N
    h = L.Handle(Lt, N, nt, colors, True, 1, 2, -1)
    h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    g = np.random.default_rng(1)
    uv = np.asfortranarray(g.standard_normal((Lt, N, 2)) + 1j * g.standard_normal((Lt, N, 2)))
    a, b, c = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, uv)
    h.call("smoqy_matvec_v", L.OP_M, b, a)
    h.call("smoqy_matvec_v", L.OP_MT, c, a)
    Mv, Mtu = h.vec_download(b), h.vec_download(c)
    u, v = uv[:, :, 0], uv[:, :, 1]
    lhs, rhs = np.vdot(u, Mv[:, :, 1]), np.vdot(Mtu[:, :, 0], v)
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)
    # fused MᵀM == Mᵀ after M;  fused MMᵀ == M after Mᵀ
    h.call("smoqy_matvec_v", L.OP_MT, c, b)
    two_pass = h.vec_download(c)
    h.call("smoqy_matvec_v", L.OP_MTM, c, a)
    assert relerr(h.vec_download(c), two_pass) < 1e-13
    h.call("smoqy_matvec_v", L.OP_MT, b, a)
    h.call("smoqy_matvec_v", L.OP_M, c, b)
    two_pass = h.vec_download(c)
    h.call("smoqy_matvec_v", L.OP_MMT, c, a)
    assert relerr(h.vec_download(c), two_pass) < 1e-13
    # oracle comparison of the fused apply (a fraction of a second even at full size)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
    o = orc.OracleFDM(nt, expV, ch, sh, True)
    assert relerr(h.vec_download(c)[:, :, 0], o.mul_MMt(u)) < 1e-13
    # FourierTransformer at this Ltau
    w = uv.copy(order="F")
    h.call("smoqy_fft_forward", L.ptr(w), 0, 2)
    assert abs(np.vdot(w, w).real - np.vdot(uv, uv).real) < 1e-12 * np.vdot(uv, uv).real
    h.call("smoqy_fft_inverse", L.ptr(w), 0, 2)
    assert relerr(w, uv) < 1e-13


@pytest.mark.parametrize("name", CONFIGS)
def test_full_size_pcg_and_action(name):
    batch = WalkerBatch(name, nwalkers=2)
    h = batch.h
    sf0 = batch.sample_pseudofermion_fields()
    sf, iters, eps = batch.calculate_fermionic_action(1e-10)
    assert np.all(iters > 0) and np.all(iters < 10000) and np.all(eps < 1e-10)
    # S_f = Φᵀ Λ⁻¹ (MᵀM)⁻¹ Λ⁻ᵀ Φ = |R|² exactly when Φ = Λᵀ Mᵀ R
    np.testing.assert_allclose(sf, sf0, rtol=5e-9)
    # independent residual check of the solve: rebuild b = Λ⁻ᵀΦ, x = ΛΨ, compare MᵀM x with b
    bvec, xvec, ax = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIVT, bvec, batch.phi)
    h.call("smoqy_lambda_apply_v", L.LAMBDA_MUL, xvec, batch.u)
    h.call("smoqy_matvec_v", L.OP_MTM, ax, xvec)
    B, AX = h.vec_download(bvec), h.vec_download(ax)
    for w in range(2):
        assert np.linalg.norm(AX[:, :, w] - B[:, :, w]) / np.linalg.norm(B[:, :, w]) < 2e-10
    # unpreconditioned solve gives the same Ψ (more iterations)
    psi_p = h.vec_download(batch.u)
    sf_n, iters_n, eps_n = batch.calculate_fermionic_action(1e-10, use_precond=False)
    assert np.all(iters_n >= iters)
    assert relerr(h.vec_download(batch.u), psi_p) < 1e-7  # both within kappa*tol of the exact solution
    np.testing.assert_allclose(sf_n, sf0, rtol=5e-9)


def test_config1_against_oracle_end_to_end():
    """configs[0] (the reference's own CPU-runnable case, L=4, Ltau=40): the whole action solve
    chain against the CPU oracle with the same random vectors."""
    name = CONFIGS[0]
    batch = WalkerBatch(name, nwalkers=1)
    m = batch.models[0]
    Lt, N = batch.Lt, batch.N
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, batch.perm, m.fpi.dtau, True)
    o = orc.OracleFDM(batch.nt, expV, ch, sh, True)
    hol = m.elph.holstein
    Lam = orc.update_lambda(Lt, N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
    g = np.random.default_rng(3)
    R = np.asfortranarray((g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N))) * np.sqrt(0.5))
    batch.h.vec_upload(batch.phi, R)
    batch.h.call("smoqy_matvec_v", L.OP_MT, batch.phi, batch.phi)
    batch.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, batch.phi, batch.phi)
    phi = orc.lambda_apply(Lam, o.mul_Mt(R), "mulT")
    assert relerr(batch.h.vec_download(batch.phi), phi) < 1e-13
    rv = np.random.default_rng(4).standard_normal(N)
    P = orc.OracleKPM(o)
    P.update(rv)
    batch.h.call("smoqy_precond_update", 0, L.ptr(rv))
    batch.h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIVT, batch.u, batch.phi)
    iters = np.zeros(1, dtype=np.int32)
    eps = np.zeros(1)
    batch.h.call("smoqy_cg_solve_v", batch.u, batch.u, C.c_double(1e-10), 10000, 1, L.ptr(iters), L.ptr(eps))
    batch.h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIV, batch.u, batch.u)
    x, ito, _ = o.cg_solve(orc.lambda_apply(Lam, phi, "ldivT"), precond=P, tol=1e-10, maxiter=10000)
    psi = orc.lambda_apply(Lam, x, "ldiv")
    assert abs(int(iters[0]) - ito) <= 2
    assert relerr(batch.h.vec_download(batch.u), psi) < 1e-8
    sf = batch.h.vec_dot(batch.phi, batch.u)[0]
    assert abs(sf - np.vdot(phi, psi)) < 1e-8 * abs(sf)


@pytest.mark.parametrize("name", ["holstein_honeycomb_L4_Ltau40", "bssh_chain_L256_Ltau200"])
def test_sweep_force_matches_oracle(name):
    """The force the synthetic sweep computes after each solve (WalkerBatch.fermionic_force) against the
    oracle's restatement of src/PFFCalculator.jl:146-155 on the same Ψ, per walker."""
    batch = WalkerBatch(name, nwalkers=2)
    batch.sample_pseudofermion_fields()
    batch.calculate_fermionic_action(1e-10)
    dS = batch.fermionic_force().copy()
    Psi = batch.h.vec_download(batch.u)
    for w, m in enumerate(batch.models):
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, batch.perm, m.fpi.dtau, True)
        o = orc.OracleFDM(batch.nt, expV, ch, sh, True)
        fc = m.force_couplings(batch.perm)
        e = orc.OracleElph(fc)
        hol = m.elph.holstein
        if hol is not None:
            Lam = orc.update_lambda(batch.Lt, batch.N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
        else:
            Lam = orc.update_lambda(batch.Lt, batch.N, m.elph.x, m.elph.dtau, [], [], [], [], [])
        P = Psi[:, :, w]
        LP = orc.lambda_apply(Lam, P, "mul")
        AP = o.mul_M(LP)
        want = orc.mul_dMdx(o, e, batch.colors, -2.0, AP, LP)
        orc.mul_dLdx(e, Lam, -2.0, o.mul_Mt(AP), P, want)
        assert np.abs(dS[w].T - want).max() < 1e-10 * np.abs(want).max()
