"""GPU tests of the host-side mirror of the reference operator API (smoqyelphqmc_amd.api):
the calls read like the reference's own call sites (src/PFFCalculator.jl, src/FermionDetMatrix.jl,
src/Measurements/GreensEstimator.jl:125-175) and are checked against the CPU oracle and the
dense known-answer matrices.  fp64 tolerances: 1e-13 for applies, rtol 1e-10 for CG solutions.
"""
import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from oracle import dense, oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def rand_vec(Lt, N, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))


@pytest.fixture(scope="module", params=[True, False], ids=["sym", "asym"])
def setup(request):
    is_sym = request.param
    m = lat.holstein_honeycomb(3, 12)
    fdm = (sq.SymFermionDetMatrix if is_sym else sq.AsymFermionDetMatrix)(m.fpi, maxiter=5000, tol=1e-10)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, is_sym)
    o = orc.OracleFDM(fdm.checkerboard_neighbor_table, expV, ch, sh, is_sym)
    return m, fdm, o, is_sym


def test_constructor_and_fields(setup):
    m, fdm, o, is_sym = setup
    assert sq.size(fdm) == (12 * 18, 12 * 18) and sq.size(fdm, 1) == 12 * 18
    assert fdm.cgs.maxiter == 5000 and fdm.cgs.tol == 1e-10
    np.testing.assert_allclose(fdm.expnΔτV, o.expV, rtol=1e-15)
    np.testing.assert_allclose(fdm.coshΔτt, o.cosh, rtol=1e-15)
    np.testing.assert_allclose(fdm.sinhΔτt, o.sinh, rtol=1e-14)
    assert len(fdm.checkerboard_colors) == 3


def test_mul_family_and_in_place(setup):
    m, fdm, o, is_sym = setup
    v = rand_vec(12, 18, 1)
    vp = np.zeros_like(v)
    for f, ref in ((sq.mul_M, o.mul_M), (sq.mul_Mt, o.mul_Mt), (sq.mul_MtM, o.mul_MtM), (sq.mul_MMt, o.mul_MMt), (sq.mul, o.mul_MtM)):
        f(vp, fdm, v)
        assert relerr(vp, ref(v)) < 1e-13
    for f, ref in ((sq.lmul_M, o.mul_M), (sq.lmul_Mt, o.mul_Mt), (sq.lmul_MtM, o.mul_MtM), (sq.lmul_MMt, o.mul_MMt), (sq.lmul, o.mul_MtM)):
        w = v.copy(order="F")
        f(fdm, w)
        assert relerr(w, ref(v)) < 1e-13
    # flat vectors are accepted like `reshaped` (src/SmoQyElPhQMC.jl:18-20)
    flat = v.ravel(order="F").copy()
    out = np.zeros_like(flat)
    sq.mul_M(out, fdm, flat)
    assert relerr(out, o.mul_M(v).ravel(order="F")) < 1e-13
    # dense known answer
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, o.expV, o.cosh, o.sinh, is_sym)
    assert relerr(out, M @ flat) < 1e-13


def test_ldiv_identity_and_kpm(setup):
    m, fdm, o, is_sym = setup
    b = rand_vec(12, 18, 2)
    x = np.zeros_like(b)
    iters, eps = sq.ldiv(x, fdm, b, preconditioner=sq.I, tol=1e-12, maxiter=5000)
    xo, ito, _ = o.cg_solve(b, tol=1e-12, maxiter=5000)
    assert abs(iters - ito) <= 2 and eps < 1e-12 and relerr(x, xo) < 1e-10
    # in-place form ldiv!(fdm, v) == `x === b`
    y = b.copy(order="F")
    it2, _ = sq.ldiv(fdm, y, tol=1e-12)
    assert it2 == iters and relerr(y, x) < 1e-12
    # dense solve
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, o.expV, o.cosh, o.sinh, is_sym)
    want = np.linalg.solve(M.conj().T @ M, dense.vec(b))
    assert relerr(dense.vec(x), want) < 1e-10
    # KPM preconditioner: same solution, mirrored state
    rng = np.random.default_rng(5)
    P = sq.KPMPreconditioner(fdm, rng=rng)
    Po = orc.OracleKPM(o)
    Po.update(np.random.default_rng(5).standard_normal(18))  # same first draw as the constructor's update
    assert P.active == Po.active
    np.testing.assert_allclose(P.bounds, Po.bounds, rtol=1e-6)  # 18 sites < 20 Lanczos steps: Ritz extremes only
    z = np.zeros_like(b)
    iters_p, eps_p = sq.ldiv(z, fdm, b, preconditioner=P, rng=rng, tol=1e-12, maxiter=5000)
    assert eps_p < 1e-12 and relerr(z, x) < 1e-9
    # ldiv!(z, P, r) and the no-op update for I (KPMPreconditioner.jl:600)
    r = rand_vec(12, 18, 3)
    out = np.zeros_like(r)
    sq.ldiv(out, P, r)
    u, v = rand_vec(12, 18, 8), rand_vec(12, 18, 9)
    uo, vo = np.zeros_like(u), np.zeros_like(v)
    sq.ldiv(uo, P, u)
    sq.ldiv(vo, P, v)
    if is_sym:  # Hermitian positive definite
        assert abs(np.vdot(u, vo) - np.vdot(uo, v)) < 1e-10 * abs(np.vdot(u, vo))
    assert np.vdot(u, uo).real > 0
    assert sq.update_preconditioner(sq.I, fdm, rng) is None


def test_update_follows_path_integral(setup):
    m, fdm, o, is_sym = setup
    m2 = lat.holstein_honeycomb(3, 12, walker=7)
    sq.update(fdm, m2.fpi)
    expV, ch, sh = orc.update_fields(m2.fpi.V, m2.fpi.t, fdm.checkerboard_perm, m2.fpi.dtau, is_sym)
    o2 = orc.OracleFDM(fdm.checkerboard_neighbor_table, expV, ch, sh, is_sym)
    v = rand_vec(12, 18, 4)
    vp = np.zeros_like(v)
    sq.mul_MtM(vp, fdm, v)
    assert relerr(vp, o2.mul_MtM(v)) < 1e-13
    sq.update(fdm, m.fpi)  # restore for the other tests of this module


def test_fourier_transformer_api():
    U = sq.FourierTransformer(np.float64, 40, 6)
    v = rand_vec(40, 6, 5)
    ft = orc.OracleFT(40, 6)
    u = np.zeros_like(v)
    sq.mul(u, U, v)
    assert relerr(u, ft.forward(v)) < 1e-13
    w = v.copy(order="F")
    sq.lmul(U, w)
    assert relerr(w, u) < 1e-15
    sq.ldiv(U, w)
    assert relerr(w, v) < 1e-13
    back = np.zeros_like(v)
    sq.ldiv(back, U, u)
    assert relerr(back, v) < 1e-13
    np.testing.assert_allclose(U.θ, np.exp(-1j * np.pi * np.arange(40) / 40))


def test_lambda_api():
    m = lat.holstein_honeycomb(3, 12)
    Λ = np.zeros((12, 18), order="F")
    sq.update_Λ(Λ, m.elph)
    hol = m.elph.holstein
    want = orc.update_lambda(12, 18, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
    np.testing.assert_allclose(Λ, want, rtol=1e-15)
    v = rand_vec(12, 18, 6)
    for f, name in ((sq.mul_Λ, "mul"), (sq.ldiv_Λ, "ldiv"), (sq.mul_Λᵀ, "mulT"), (sq.ldiv_Λᵀ, "ldivT")):
        out = np.zeros_like(v)
        f(out, Λ, v)
        assert relerr(out, orc.lambda_apply(Λ, v, name)) < 1e-14
        w = v.copy(order="F")
        f(w, Λ, w)  # aliased, as relied upon at PFFCalculator.jl:73, 107
        assert relerr(w, out) < 1e-15
    # ldiv_Λ ∘ mul_Λ = I
    a, b = np.zeros_like(v), np.zeros_like(v)
    sq.mul_Lambda(a, Λ, v)
    sq.ldiv_Lambda(b, Λ, a)
    assert relerr(b, v) < 1e-14


def test_pff_calculator_action(setup):
    m, fdm, o, is_sym = setup
    pff = sq.PFFCalculator(m.elph, fdm)
    R = rand_vec(12, 18, 7) * np.sqrt(0.5)
    Sf0 = sq.sample_pseudofermion_fields(pff, m.elph, fdm, R=R)
    assert abs(Sf0 - np.vdot(R, R).real) < 1e-12 * Sf0
    hol = m.elph.holstein
    Λ = orc.update_lambda(12, 18, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
    Φ_want = orc.lambda_apply(Λ, o.mul_Mt(R), "mulT")
    assert relerr(pff.Φ, Φ_want) < 1e-13
    rng = np.random.default_rng(11)
    P = sq.KPMPreconditioner(fdm, rng=rng)
    Sf, iters, eps = sq.calculate_fermionic_action(pff, m.elph, fdm, P, rng, 1e-12, 5000)
    # S_f = Φᵀ Λ⁻¹ [MᵀM]⁻¹ Λ⁻ᵀ Φ with Φ = Λᵀ Mᵀ R  ==>  S_f = |R|²  (the heat-bath identity)
    assert abs(Sf - Sf0) < 1e-9 * Sf0
    assert 0 < iters < 5000 and eps < 1e-12
    # and against the oracle's own solve
    psi0 = orc.lambda_apply(Λ, Φ_want, "ldivT")
    xo, _, _ = o.cg_solve(psi0, tol=1e-13, maxiter=5000)
    Ψ_want = orc.lambda_apply(Λ, xo, "ldiv")
    assert relerr(pff.u, Ψ_want) < 1e-9
    np.testing.assert_allclose(pff.Λ, Λ, rtol=1e-15)
    # sampling with the rng: |R|² ~ Lt*N on average
    s = sq.sample_pseudofermion_fields(pff, m.elph, fdm, rng)
    assert 0.5 * 12 * 18 < s < 1.5 * 12 * 18


def test_multi_rhs_greens_estimator_style():
    """update_greens_estimator! (src/Measurements/GreensEstimator.jl:125-175): Nrv unit-modulus
    random vectors, GR = M⁻¹R through MᵀM x = MᵀR, all right-hand sides in one batched solve."""
    m = lat.holstein_honeycomb(3, 12)
    Nrv = 5
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-10, nrhs=Nrv)
    h = fdm.handle
    g = np.random.default_rng(12)
    R = g.standard_normal((12, 18, Nrv)) + 1j * g.standard_normal((12, 18, Nrv))
    R = np.asfortranarray(R / np.abs(R))  # :141-142
    rid, mtr, gr = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(rid, R)
    h.call("smoqy_matvec_v", 1, mtr, rid)  # mul_Mt!  :157
    iters = np.zeros(Nrv, dtype=np.int32)
    eps = np.zeros(Nrv)
    import ctypes as C

    h.call("smoqy_cg_solve_v", gr, mtr, C.c_double(1e-12), 5000, 0, iters.ctypes.data_as(C.c_void_p), eps.ctypes.data_as(C.c_void_p))  # :159-166 (warm start from GR = 0)
    GR = h.vec_download(gr)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    for n in range(Nrv):
        want = np.linalg.solve(M, dense.vec(R[:, :, n]))
        assert relerr(dense.vec(GR[:, :, n]), want) < 1e-9
    assert np.all(iters > 0) and np.all(eps < 1e-12)


def test_errors_surface_as_exceptions(setup):
    m, fdm, o, is_sym = setup
    with pytest.raises(ValueError):
        sq.mul_M(np.zeros((3, 3), dtype=complex), fdm, np.zeros((12, 18), dtype=complex))
    with pytest.raises(TypeError):
        sq.cg_solve(np.zeros(4, dtype=complex), np.eye(4), np.zeros(4, dtype=complex))
    # NaN right-hand side: the C ABI reports a non-finite residual, the shim raises, and the
    # reference's try/catch would reject the update (src/EFAPFFHMCUpdater.jl:168-187)
    b = rand_vec(12, 18, 13)
    b[0, 0] = np.nan
    with pytest.raises(sq.api.L.SmoqyError):
        sq.ldiv(np.zeros_like(b), fdm, b, tol=1e-10, maxiter=50)


def test_checkerboard_entry_points(setup):
    """checkerboard_lmul!/ldiv!/mul! with `transposed` and a colour `interval`
    (src/checkerboard_matrix_multiply.jl:2-145) against the oracle."""
    m, fdm, o, is_sym = setup
    v = rand_vec(12, 18, 21)
    for tr in (False, True):
        w = v.copy(order="F")
        sq.checkerboard_lmul(w, fdm, transposed=tr)
        assert relerr(w, o.checkerboard(v, transposed=tr)) < 1e-14
        sq.checkerboard_ldiv(w, fdm, transposed=tr)  # inverse of the same product
        assert relerr(w, v) < 1e-13
        w = v.copy(order="F")
        sq.checkerboard_ldiv(w, fdm, transposed=tr)
        assert relerr(w, o.checkerboard(v, transposed=tr, inverse=True)) < 1e-14
        for col in fdm.checkerboard_colors:  # one colour at a time, as the force code does
            w = v.copy(order="F")
            sq.checkerboard_lmul(w, fdm, transposed=tr, interval=col)
            assert relerr(w, o.checkerboard(v, transposed=tr, interval=(col.start - 1, col.stop - 1))) < 1e-14
    out = np.zeros_like(v)
    sq.checkerboard_mul(out, v, fdm, transposed=True)
    assert relerr(out, o.checkerboard(v, transposed=True)) < 1e-14


@pytest.mark.parametrize("is_sym", [True, False])
def test_kpm_ldiv_on_a_real_vector(is_sym):
    """The real-vector ldiv! methods (src/KPMPreconditioner.jl:288-352, 417-485) through the mirror: the device evaluates half the
    frequencies and mirrors the rest; compared with the complex method on the same (promoted) vector.  Oracle parity of the
    device method itself: tests/test_gpu_parity.py::test_kpm_real_vector_apply."""
    m = lat.holstein_honeycomb(4, 10)  # N = 32 > 20 Lanczos steps
    fdm = (sq.SymFermionDetMatrix if is_sym else sq.AsymFermionDetMatrix)(m.fpi, maxiter=5000, tol=1e-10)
    P = sq.KPMPreconditioner(fdm, rng=np.random.default_rng(5))
    assert P.active
    g = np.random.default_rng(6)
    u = np.asfortranarray(g.standard_normal((10, 32)))
    up = np.zeros_like(u)
    sq.ldiv(up, P, u)
    uc = np.asfortranarray(u.astype(complex))
    upc = np.zeros_like(uc)
    sq.ldiv(upc, P, uc)
    assert np.abs(up - upc.real).max() < 1e-13 * np.abs(upc).max()
    assert np.abs(upc.imag).max() < 1e-12 * np.abs(upc.real).max()  # the complex method keeps a real vector real (Sym and Asym)
