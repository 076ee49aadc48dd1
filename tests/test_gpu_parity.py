"""GPU parity tests (run with ``-m gpu`` on the MI355X box): every C-ABI entry point of
libsmoqy_hip.so against the CPU oracle on identical seeded inputs.

Tolerances (fp64): single operator applies 1e-13 relative to the vector's max norm (the device
uses fused multiply-adds and a different summation order than the oracle); CG solutions
rtol 1e-10 as BASELINE.json states.
"""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice

OP_TOL = 1e-13


def model_of(kind, walker=0):
    if kind == "honeycomb":
        return lat.holstein_honeycomb(3, 10, walker=walker)
    if kind == "square":
        return lat.ossh_square(4, 7, walker=walker)
    if kind == "chain":
        return lat.bssh_chain(10, 9, walker=walker)
    if kind == "honeycomb_L4":
        return lat.holstein_honeycomb(4, 40, walker=walker)
    if kind == "square_L6":
        return lat.ossh_square(6, 12, walker=walker)
    if kind == "chain_L24":
        return lat.bssh_chain(24, 16, walker=walker)
    if kind == "chain_odd":
        return lat.bssh_chain(24, 9, walker=walker)
    raise KeyError(kind)


class Problem:
    def __init__(self, kind, is_sym=True, nwalkers=1, nrhs=1, smooth=False):
        self.models = [model_of(kind, w) for w in range(nwalkers)]
        m0 = self.models[0]
        self.nt, self.perm, self.colors = lat.checkerboard_decomposition(m0.fpi.neighbor_table)
        self.Lt, self.N = m0.fpi.Ltau, m0.fpi.N
        self.is_sym = is_sym
        self.h = L.Handle(self.Lt, self.N, self.nt, self.colors, is_sym, nwalkers, nrhs)
        self.oracles = []
        for w, m in enumerate(self.models):
            expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, self.perm, m.fpi.dtau, is_sym)
            self.h.call("smoqy_update_fields", w, L.ptr(expV), L.ptr(ch), L.ptr(sh))
            self.oracles.append(orc.OracleFDM(self.nt, expV, ch, sh, is_sym))

    def rand(self, count, seed):
        g = np.random.default_rng(seed)
        shape = (self.Lt, self.N, count)
        return np.asfortranarray(g.standard_normal(shape) + 1j * g.standard_normal(shape))


def relerr(got, want):
    return np.abs(got - want).max() / np.abs(want).max()


def test_upload_download_roundtrip():
    p = Problem("square", nwalkers=2, nrhs=3)
    v = p.rand(6, 1)
    vid = p.h.vec_alloc()
    p.h.vec_upload(vid, v)
    np.testing.assert_array_equal(p.h.vec_download(vid), v)
    # partial ranges
    w = p.rand(2, 2)
    p.h.vec_upload(vid, w, sys0=3, count=2)
    back = p.h.vec_download(vid)
    np.testing.assert_array_equal(back[:, :, 3:5], w)
    np.testing.assert_array_equal(back[:, :, :3], v[:, :, :3])
    np.testing.assert_array_equal(p.h.vec_download(vid, 5, 1), v[:, :, 5])


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("Tc", [1, 2, 3, 4])
@pytest.mark.parametrize("generic", [False, True])
def test_matvec_all_ops(kind, is_sym, Tc, generic):
    p = Problem(kind, is_sym)
    p.h.call("smoqy_matvec_force_generic", int(generic))
    p.h.call("smoqy_set_tau_chunk", Tc)
    v = p.rand(1, 3)
    o = p.oracles[0]
    for op, fn in ((L.OP_M, o.mul_M), (L.OP_MT, o.mul_Mt), (L.OP_MTM, o.mul_MtM), (L.OP_MMT, o.mul_MMt)):
        out = np.zeros_like(v)
        p.h.call("smoqy_matvec", op, L.ptr(out), L.ptr(v), 0, 1)
        assert relerr(out[:, :, 0], fn(v[:, :, 0])) < OP_TOL, (kind, is_sym, Tc, op)


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain"])
@pytest.mark.parametrize("Tc", [1, 2])
def test_matvec_large_batch_uses_the_lds_resident_kernel(kind, Tc):
    """More than 8 systems per launch dispatch the Sym operator to the LDS-resident kernel (kernels_fdm_fast.hip),
    up to 8 to the owner-computes kernel (kernels_fdm_own.hip): both against the oracle, and against each other."""
    p = Problem(kind, True, nwalkers=3, nrhs=4)  # 12 systems
    p.h.call("smoqy_set_tau_chunk", Tc)
    v = p.rand(12, 8)
    a, b = p.h.vec_alloc(), p.h.vec_alloc()
    p.h.vec_upload(a, v)
    for op, name in ((L.OP_M, "mul_M"), (L.OP_MT, "mul_Mt"), (L.OP_MTM, "mul_MtM"), (L.OP_MMT, "mul_MMt")):
        p.h.call("smoqy_matvec_v", op, b, a)
        big = p.h.vec_download(b)
        for s in (0, 5, 11):
            assert relerr(big[:, :, s], getattr(p.oracles[s // 4], name)(v[:, :, s])) < OP_TOL, (kind, Tc, name, s)
        small = np.zeros((p.Lt, p.N, 2), dtype=complex, order="F")  # 2 systems: owner-computes kernel
        p.h.call("smoqy_matvec", op, L.ptr(small), L.ptr(np.asfortranarray(v[:, :, 6:8])), 6, 2)
        assert relerr(small, big[:, :, 6:8]) < 4e-15


def test_matvec_in_place_and_batched():
    p = Problem("honeycomb", True, nwalkers=2, nrhs=2)
    v = p.rand(4, 4)
    a, b = p.h.vec_alloc(), p.h.vec_alloc()
    p.h.vec_upload(a, v)
    p.h.call("smoqy_matvec_v", L.OP_M, b, a)
    p.h.call("smoqy_matvec_v", L.OP_MT, a, a)  # in place (lmul_Mt!)
    got_M, got_Mt = p.h.vec_download(b), p.h.vec_download(a)
    for s in range(4):
        o = p.oracles[s // 2]
        assert relerr(got_M[:, :, s], o.mul_M(v[:, :, s])) < OP_TOL
        assert relerr(got_Mt[:, :, s], o.mul_Mt(v[:, :, s])) < OP_TOL
    # host form on a sub-range uses the right walker's fields
    out = np.zeros((p.Lt, p.N, 1), dtype=complex, order="F")
    p.h.call("smoqy_matvec", L.OP_MTM, L.ptr(out), L.ptr(np.asfortranarray(v[:, :, 3:4])), 3, 1)
    assert relerr(out[:, :, 0], p.oracles[1].mul_MtM(v[:, :, 3])) < OP_TOL
    # dot
    p.h.vec_upload(a, v)
    d = p.h.vec_dot(a, b)
    for s in range(4):
        want = np.vdot(v[:, :, s], got_M[:, :, s])
        assert abs(d[s] - want) < 1e-12 * abs(want)


@pytest.mark.parametrize("is_sym", [True, False])
def test_update_from_path_integral_and_get_fields(is_sym):
    p = Problem("square", is_sym)
    m = p.models[0]
    m.elph.x[...] *= 1.7
    m.refresh_from_x()
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, p.perm, m.fpi.dtau, is_sym)
    p.h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(p.perm), C.c_double(m.fpi.dtau))
    g_e, g_c, g_s = np.zeros_like(expV), np.zeros_like(ch), np.zeros_like(sh)
    p.h.call("smoqy_get_fields", 0, L.ptr(g_e), L.ptr(g_c), L.ptr(g_s))
    np.testing.assert_allclose(g_e, expV, rtol=1e-15)
    np.testing.assert_allclose(g_c, ch, rtol=1e-15)
    np.testing.assert_allclose(g_s, sh, rtol=1e-14)


def test_lambda_ops():
    p = Problem("honeycomb", True, nwalkers=2, nrhs=1)
    v = p.rand(2, 5)
    vid, wid = p.h.vec_alloc(), p.h.vec_alloc()
    lams = []
    for w, m in enumerate(p.models):
        hol = m.elph.holstein
        lam = orc.update_lambda(p.Lt, p.N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
        lams.append(lam)
        p.h.call("smoqy_lambda_update", w, L.ptr(m.elph.x), m.elph.x.shape[0], C.c_double(m.elph.dtau), len(hol.alpha), L.ptr(np.asarray(hol.coupling_to_phonon, dtype=np.int64)),
                 L.ptr(np.asarray(hol.coupling_to_site, dtype=np.int64)), L.ptr(np.asarray(hol.alpha, dtype=np.float64)), L.ptr(np.asarray(hol.alpha3, dtype=np.float64)),
                 L.ptr(np.asarray(hol.ph_sym_form, dtype=np.int32)))
        got = np.zeros_like(lam)
        p.h.call("smoqy_lambda_get", w, L.ptr(got))
        np.testing.assert_allclose(got, lam, rtol=1e-15)
    for name, op in orc.LAMBDA_OPS.items():
        p.h.vec_upload(vid, v)
        p.h.call("smoqy_lambda_apply_v", op, wid, vid)
        p.h.call("smoqy_lambda_apply_v", op, vid, vid)  # aliased form relied on by PFFCalculator.jl:73, 107
        out, out2 = p.h.vec_download(wid), p.h.vec_download(vid)
        for s in range(2):
            want = orc.lambda_apply(lams[s], v[:, :, s], name)
            assert relerr(out[:, :, s], want) < 1e-15 * 10
            assert relerr(out2[:, :, s], want) < 1e-15 * 10
    # host form with caller-supplied Λ
    out = np.zeros((p.Lt, p.N, 1), dtype=complex, order="F")
    p.h.call("smoqy_lambda_apply", L.LAMBDA_LDIVT, L.ptr(out), L.ptr(np.asfortranarray(v[:, :, 1:2])), L.ptr(lams[0]), 1, 1)
    assert relerr(out[:, :, 0], orc.lambda_apply(lams[0], v[:, :, 1], "ldivT")) < 1e-14


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain", "honeycomb_L4", "square_L6", "chain_L24"])  # Ltau = 10, 7, 9, 40, 12, 16
@pytest.mark.parametrize("rocfft", [False, True])
def test_fourier_transformer(kind, rocfft):
    p = Problem(kind, True, nwalkers=1, nrhs=2)
    p.h.call("smoqy_fft_use_rocfft", int(rocfft))  # own Stockham kernels vs the rocFFT plans
    v = p.rand(2, 6)
    ft = orc.OracleFT(p.Lt, p.N)
    w = v.copy(order="F")
    p.h.call("smoqy_fft_forward", L.ptr(w), 0, 2)
    for s in range(2):
        assert relerr(w[:, :, s], ft.forward(v[:, :, s])) < 1e-13
    p.h.call("smoqy_fft_inverse", L.ptr(w), 0, 2)
    assert relerr(w, v) < 1e-13
    # unitarity
    vid = p.h.vec_alloc()
    p.h.vec_upload(vid, v)
    n0 = p.h.vec_dot(vid, vid)
    p.h.call("smoqy_fft_forward_v", vid)
    n1 = p.h.vec_dot(vid, vid)
    np.testing.assert_allclose(n1.real, n0.real, rtol=1e-13)


@pytest.mark.parametrize("kind", ["honeycomb", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_cg_unpreconditioned(kind, is_sym):
    p = Problem(kind, is_sym, nwalkers=1, nrhs=2)
    b = p.rand(2, 7)
    x = np.zeros_like(b)
    iters = np.zeros(2, dtype=np.int32)
    eps = np.zeros(2)
    p.h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, 2, C.c_double(1e-12), 5000, 0, L.ptr(iters), L.ptr(eps))
    for s in range(2):
        xo, ito, epo = p.oracles[0].cg_solve(b[:, :, s], tol=1e-12, maxiter=5000)
        assert relerr(x[:, :, s], xo) < 1e-10
        assert abs(int(iters[s]) - ito) <= 2 and eps[s] < 1e-12
        assert relerr(p.oracles[0].mul_MtM(x[:, :, s]), b[:, :, s]) < 1e-11
    # warm start from the solution: 0 iterations (ConjugateGradient.jl:133)
    p.h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 0, 0, 2, C.c_double(1e-10), 5000, 0, L.ptr(iters), L.ptr(eps))
    assert iters.tolist() == [0, 0]
    # maxiter reached is not an error: returns (maxiter, eps)
    x2 = np.zeros_like(b)
    p.h.call("smoqy_cg_solve", L.ptr(x2), L.ptr(b), 1, 0, 2, C.c_double(1e-14), 3, 0, L.ptr(iters), L.ptr(eps))
    assert iters.tolist() == [3, 3] and np.all(eps > 1e-14)


def _precond_state(p, w=0):
    act = C.c_int(0)
    bounds = np.zeros(2)
    order = np.zeros(p.Lt, dtype=np.int32)
    norder = C.c_int(0)
    la, lb = np.zeros(20), np.zeros(19)
    p.h.call("smoqy_precond_get", w, C.byref(act), bounds.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(norder),
             la.ctypes.data_as(C.POINTER(C.c_double)), lb.ctypes.data_as(C.POINTER(C.c_double)))
    return bool(act.value), bounds, order[: norder.value], la, lb


@pytest.mark.parametrize("kind", ["honeycomb_L4", "square_L6", "chain_L24"])  # N > 20 Lanczos steps
@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("generic", [False, True])
def test_kpm_preconditioner_state_and_apply(kind, is_sym, generic):
    p = Problem(kind, is_sym, nwalkers=1, nrhs=2)
    p.h.call("smoqy_precond_force_generic", int(generic))
    rv = np.random.default_rng(8).standard_normal(p.N)
    P = orc.OracleKPM(p.oracles[0])
    P.update(rv)
    p.h.call("smoqy_precond_update", 0, L.ptr(rv))
    act, bounds, order, la, lb = _precond_state(p)
    oa, ob = P.lanczos()
    assert act == P.active
    # Lanczos amplifies rounding differences step by step (loss of orthogonality): the early
    # coefficients agree tightly, the Ritz extremes (what the preconditioner uses) stay stable
    np.testing.assert_allclose(la[:8], oa[:8], rtol=1e-10)
    np.testing.assert_allclose(lb[:8], ob[:8], rtol=1e-9)
    np.testing.assert_allclose(la, oa, rtol=1e-5)
    np.testing.assert_allclose(bounds, P.bounds, rtol=1e-9)
    assert np.array_equal(order, P.order)
    for slot in (0, len(order) - 1):
        c = np.zeros(int(order[slot]), dtype=complex)
        p.h.call("smoqy_precond_get_coefs", 0, slot, L.ptr(c))
        np.testing.assert_allclose(c, P.coefs(slot), rtol=1e-9, atol=1e-12)
    v = p.rand(2, 9)
    out = np.zeros_like(v)
    p.h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, 2)
    for s in range(2):
        assert relerr(out[:, :, s], P.apply(v[:, :, s])) < 1e-11


@pytest.mark.parametrize("kind", ["honeycomb_L4", "square_L6", "chain_odd"])  # Ltau = 40, 12, 9 (odd: the middle frequency mirrors itself)
@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("generic", [False, True])
def test_kpm_real_vector_apply(kind, is_sym, generic):
    """smoqy_precond_apply_real against the oracle's restatement of the real-vector ldiv! methods
    (src/KPMPreconditioner.jl:288-352 Sym, :417-485 Asym): half the frequencies, conjugate mirror, real part."""
    p = Problem(kind, is_sym, nwalkers=1, nrhs=2)
    p.h.call("smoqy_precond_force_generic", int(generic))
    rv = np.random.default_rng(8).standard_normal(p.N)
    P = orc.OracleKPM(p.oracles[0])
    P.update(rv)
    p.h.call("smoqy_precond_update", 0, L.ptr(rv))
    assert P.active
    u = np.asfortranarray(np.random.default_rng(19).standard_normal((p.Lt, p.N, 2)))
    out = np.zeros_like(u)
    p.h.call("smoqy_precond_apply_real", L.ptr(out), L.ptr(u), 0, 2)
    for s in range(2):
        want = P.apply_real(u[:, :, s])
        assert np.abs(out[:, :, s] - want).max() < 1e-11 * np.abs(want).max()
    # a sub-range call touches only its own system
    one = np.zeros((p.Lt, p.N, 1), order="F")
    p.h.call("smoqy_precond_apply_real", L.ptr(one), L.ptr(np.asfortranarray(u[:, :, 1:2])), 1, 1)
    assert np.abs(one[:, :, 0] - out[:, :, 1]).max() < 1e-13 * np.abs(out).max()


@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("rocfft", [False, True])
def test_cg_preconditioned(is_sym, graph, rocfft):
    p = Problem("honeycomb_L4", is_sym, nwalkers=2, nrhs=2)
    p.h.call("smoqy_fft_use_rocfft", int(rocfft))  # fused tau-FFT iteration vs the rocFFT + BLAS-1 kernels
    p.h.call("smoqy_cg_use_graph", int(graph))  # hipGraph replay of the captured iteration must not change anything
    b = p.rand(4, 10)
    Ps = []
    for w in range(2):
        rv = np.random.default_rng(11 + w).standard_normal(p.N)
        P = orc.OracleKPM(p.oracles[w])
        P.update(rv)
        p.h.call("smoqy_precond_update", w, L.ptr(rv))
        assert P.active
        Ps.append(P)
    x = np.zeros_like(b)
    iters = np.zeros(4, dtype=np.int32)
    eps = np.zeros(4)
    p.h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, 4, C.c_double(1e-10), 10000, 1, L.ptr(iters), L.ptr(eps))
    for s in range(4):
        o = p.oracles[s // 2]
        xo, ito, epo = o.cg_solve(b[:, :, s], precond=Ps[s // 2], tol=1e-10, maxiter=10000)
        x_tight, _, _ = o.cg_solve(b[:, :, s], precond=Ps[s // 2], tol=1e-14, maxiter=10000)
        assert abs(int(iters[s]) - ito) <= 2, (iters, ito)
        assert eps[s] < 1e-10
        # same iterate as the oracle, and both within kappa*tol of the exact solution
        assert relerr(x[:, :, s], xo) < 1e-9
        assert relerr(x[:, :, s], x_tight) < 1e-8
        assert relerr(o.mul_MtM(x[:, :, s]), b[:, :, s]) < 1e-9


def test_error_reporting():
    m = model_of("chain")
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    bad = colors.copy()
    bad[1, 0] += 1  # overlapping colours
    with pytest.raises(L.SmoqyError):
        L.Handle(m.fpi.Ltau, m.fpi.N, nt, bad)
    h = L.Handle(m.fpi.Ltau, m.fpi.N, nt, colors)
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_matvec_v", 0, 5, 6)  # unknown vector ids
