"""Matrix-element type T = ComplexF64 (complex hoppings) on the GPU against the CPU oracle, whose complex-T restatement is pinned by
dense matrices in tests/test_oracle_complex_T.py.  Reference: the bond factor [[c, s], [conj(s), c]] with
s = sign(conj t)·sinh(Δτ′|t|) (src/checkerboard_matrix_multiply.jl:60-68, src/FermionDetMatrix.jl:224-231); the operator type is
FermionDetMatrix{T<:Number} (:19).  Round 4: Sym complex handles run the register-resident operator kernel (fdm_fast_kernel<…, CPLX>) up to
a τ-chunk of 2 and the register-resident Chebyshev kernel (cheb_fast_kernel<true, NCOL, CPLX>); Asym complex handles, larger chunks and
the Lanczos bounds run on the generic kernels (kernels_fdm.hip, cheb_generic_kernel, generic Lanczos) — both families are compared with the oracle here."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice
OP_TOL = 1e-13


def relerr(got, want):
    return np.abs(got - want).max() / np.abs(want).max()


def complex_problem(kind, is_sym, nwalkers=1, nrhs=1):
    models, ts = [], []
    for w in range(nwalkers):
        if kind == "honeycomb":
            m = lat.holstein_honeycomb(4, 40, walker=w)
        elif kind == "square":
            m = lat.ossh_square(6, 12, walker=w)
        else:
            m = lat.bssh_chain(24, 9, walker=w)
        g = np.random.default_rng(300 + w)
        Nh, Lt = m.fpi.t.shape
        ts.append(np.asfortranarray(m.fpi.t * np.exp(1j * g.uniform(0, 2 * np.pi, Nh))[:, None]))
        models.append(m)
    m0 = models[0]
    nt, perm, colors = lat.checkerboard_decomposition(m0.fpi.neighbor_table)
    Lt, N = m0.fpi.Ltau, m0.fpi.N
    h = L.Handle(Lt, N, nt, colors, is_sym, nwalkers, nrhs, is_complex=True)
    oracles = []
    for w, (m, t) in enumerate(zip(models, ts)):
        expV, ch, sh = orc.update_fields(m.fpi.V, t, perm, m.fpi.dtau, is_sym)
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(t), L.ptr(perm), C.c_double(m.fpi.dtau))
        oracles.append(orc.OracleFDM(nt, expV, ch, sh, is_sym))
    return h, oracles, (Lt, N, nt, perm, colors), models, ts


def rand(Lt, N, count, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N, count)) + 1j * g.standard_normal((Lt, N, count)))


@pytest.mark.parametrize("is_sym", [True, False])
def test_fields_complex_T(is_sym):
    h, o, (Lt, N, nt, perm, colors), models, ts = complex_problem("square", is_sym)
    Nh = nt.shape[1]
    e = np.zeros((Lt, N), order="F")
    c = np.zeros((Lt, Nh), order="F", dtype=complex)
    s = np.zeros((Lt, Nh), order="F", dtype=complex)
    h.call("smoqy_get_fields", 0, L.ptr(e), L.ptr(c), L.ptr(s))
    np.testing.assert_allclose(e, o[0].expV, rtol=1e-15)
    np.testing.assert_allclose(c.real, o[0].cosh, rtol=1e-15)
    assert np.all(c.imag == 0)
    np.testing.assert_allclose(s, o[0].sinh + 1j * o[0].sinh_im, rtol=1e-13, atol=1e-17)
    # smoqy_update_fields takes the complex arrays back (round trip through the other entry point)
    h2 = L.Handle(Lt, N, nt, colors, is_sym, 1, 1, is_complex=True)
    h2.call("smoqy_update_fields", 0, L.ptr(e), L.ptr(c), L.ptr(s))
    v = rand(Lt, N, 1, 2)
    a, b = np.zeros_like(v), np.zeros_like(v)
    h.call("smoqy_matvec", L.OP_MTM, L.ptr(a), L.ptr(v), 0, 1)
    h2.call("smoqy_matvec", L.OP_MTM, L.ptr(b), L.ptr(v), 0, 1)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("Tc", [1, 2, 3])
def test_matvec_complex_T(kind, is_sym, Tc):
    h, o, (Lt, N, *_), *_ = complex_problem(kind, is_sym, nwalkers=2, nrhs=2)
    h.call("smoqy_set_tau_chunk", Tc)
    v = rand(Lt, N, 4, 3)
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    for op, name in ((L.OP_M, "mul_M"), (L.OP_MT, "mul_Mt"), (L.OP_MTM, "mul_MtM"), (L.OP_MMT, "mul_MMt")):
        h.call("smoqy_matvec_v", op, b, a)
        got = h.vec_download(b)
        for s in range(4):
            assert relerr(got[:, :, s], getattr(o[s // 2], name)(v[:, :, s])) < OP_TOL, (kind, is_sym, Tc, name, s)
    h.call("smoqy_matvec_v", L.OP_MTM, b, a)
    assert h.describe()["mtm"] == ("fdm_fast_kernel" if is_sym and Tc <= 2 else "fdm_kernel"), h.describe()
    # adjoint identity across separate launches
    h.call("smoqy_matvec_v", L.OP_M, b, a)
    Mv = h.vec_download(b)
    h.call("smoqy_matvec_v", L.OP_MT, b, a)
    Mtv = h.vec_download(b)
    lhs, rhs = np.vdot(v[:, :, 1], Mv[:, :, 0] * 0 + Mv[:, :, 1]), np.vdot(Mtv[:, :, 1], v[:, :, 1])
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


@pytest.mark.parametrize("is_sym", [True, False])
def test_checkerboard_complex_T(is_sym):
    h, o, (Lt, N, nt, perm, colors), *_ = complex_problem("square", is_sym)
    v = rand(Lt, N, 1, 4)
    ncol = colors.shape[1]
    for tr in (False, True):
        for inv in (False, True):
            w = v.copy(order="F")
            h.call("smoqy_checkerboard", L.ptr(w), int(inv), int(tr), 0, ncol, 0, 1)
            assert relerr(w[:, :, 0], o[0].checkerboard(v[:, :, 0], transposed=tr, inverse=inv)) < 1e-14
        for col in range(ncol):  # one colour at a time, as the force code walks them
            w = v.copy(order="F")
            h.call("smoqy_checkerboard", L.ptr(w), 0, int(tr), col, 1, 0, 1)
            want = o[0].checkerboard(v[:, :, 0], transposed=tr, interval=(int(colors[0, col]) - 1, int(colors[1, col])))
            assert relerr(w[:, :, 0], want) < 1e-14


@pytest.mark.parametrize("kind", ["honeycomb", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_cg_unpreconditioned_complex_T(kind, is_sym):
    h, o, (Lt, N, *_), *_ = complex_problem(kind, is_sym, nwalkers=1, nrhs=2)
    b = rand(Lt, N, 2, 7)
    x = np.zeros_like(b)
    iters = np.zeros(2, dtype=np.int32)
    eps = np.zeros(2)
    h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, 2, C.c_double(1e-12), 5000, 0, L.ptr(iters), L.ptr(eps))
    for s in range(2):
        xo, ito, epo = o[0].cg_solve(b[:, :, s], tol=1e-12, maxiter=5000)
        assert relerr(x[:, :, s], xo) < 1e-10
        assert abs(int(iters[s]) - ito) <= 2 and eps[s] < 1e-12
        assert relerr(o[0].mul_MtM(x[:, :, s]), b[:, :, s]) < 1e-11


def _precond_state(h, Lt):
    act, norder = C.c_int(0), C.c_int(0)
    bounds, order = np.zeros(2), np.zeros(Lt, dtype=np.int32)
    la, lb = np.zeros(20), np.zeros(19)
    h.call("smoqy_precond_get", 0, C.byref(act), bounds.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(norder),
           la.ctypes.data_as(C.POINTER(C.c_double)), lb.ctypes.data_as(C.POINTER(C.c_double)))
    return bool(act.value), bounds, order[: norder.value], la, lb


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_kpm_preconditioner_and_pcg_complex_T(kind, is_sym):
    h, o, (Lt, N, *_), *_ = complex_problem(kind, is_sym, nwalkers=1, nrhs=2)
    g = np.random.default_rng(8)
    rv = np.ascontiguousarray((g.standard_normal(N) + 1j * g.standard_normal(N)) * np.sqrt(0.5))  # randn! on a Vector{ComplexF64}
    P = orc.OracleKPM(o[0])
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    act, bounds, order, la, lb = _precond_state(h, Lt)
    oa, ob = P.lanczos()
    assert act == P.active and P.active
    np.testing.assert_allclose(la[:8], oa[:8], rtol=1e-10)
    np.testing.assert_allclose(lb[:8], ob[:8], rtol=1e-9)
    np.testing.assert_allclose(bounds, P.bounds, rtol=1e-9)
    assert np.array_equal(order, P.order)
    v = rand(Lt, N, 2, 9)
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, 2)
    for s in range(2):
        assert relerr(out[:, :, s], P.apply(v[:, :, s])) < 1e-11
    # Sym: the register-resident Chebyshev kernel with the complex bond factor (round 4); Asym: the generic kernel.  Both forms of the Sym
    # handle must agree with the oracle (smoqy_precond_force_generic switches)
    assert h.describe()["cheb"] == ("cheb_fast_kernel<complex T>" if is_sym else "cheb_generic_kernel"), h.describe()
    if is_sym:
        h.call("smoqy_precond_force_generic", 1)
        out2 = np.zeros_like(v)
        h.call("smoqy_precond_apply", L.ptr(out2), L.ptr(v), 0, 2)
        assert h.describe()["cheb"] == "cheb_generic_kernel"
        assert relerr(out2, out) < 1e-12
        h.call("smoqy_precond_force_generic", 0)
    for rocfft in (0, 1):  # fused tau-FFT iteration and the rocFFT + BLAS-1 form
        h.call("smoqy_fft_use_rocfft", rocfft)
        x = np.zeros_like(v)
        iters = np.zeros(2, dtype=np.int32)
        eps = np.zeros(2)
        h.call("smoqy_cg_solve", L.ptr(x), L.ptr(v), 1, 0, 2, C.c_double(1e-10), 10000, 1, L.ptr(iters), L.ptr(eps))
        for s in range(2):
            xo, ito, epo = o[0].cg_solve(v[:, :, s], precond=P, tol=1e-10, maxiter=10000)
            assert abs(int(iters[s]) - ito) <= 2 and eps[s] < 1e-10, (iters, ito)
            assert relerr(x[:, :, s], xo) < 1e-9
            assert relerr(o[0].mul_MtM(x[:, :, s]), v[:, :, s]) < 1e-9


def test_mirror_with_complex_hoppings():
    """The reference-shaped mirror: a FermionPathIntegral with complex t makes a FermionDetMatrix{ComplexF64}."""
    m = lat.holstein_honeycomb(4, 10)
    g = np.random.default_rng(1)
    m.fpi.t = np.asfortranarray(m.fpi.t * np.exp(1j * g.uniform(0, 2 * np.pi, m.fpi.t.shape[0]))[:, None])
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-10)
    assert sq.eltype(fdm) is np.complex128 and np.iscomplexobj(fdm.sinhΔτt)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
    o = orc.OracleFDM(nt, expV, ch, sh, True)
    v = rand(10, 32, 1, 5)[:, :, 0].copy(order="F")
    out = np.zeros_like(v)
    sq.mul_MtM(out, fdm, v)
    assert relerr(out, o.mul_MtM(v)) < OP_TOL
    P = sq.KPMPreconditioner(fdm, rng=np.random.default_rng(5))
    assert P.active
    x = np.zeros_like(v)
    iters, eps = sq.ldiv(x, fdm, v, preconditioner=P, rng=np.random.default_rng(6))
    assert eps < 1e-10 and relerr(o.mul_MtM(x), v) < 1e-9


def _flux_ssh(kind, seed=0):
    """An SSH model threaded by a static flux (tests/test_oracle_complex_T.py::flux_ssh_model): complex bare hoppings e^{iθ_h} and complex
    couplings α e^{iθ_h}, so that t_h = (1 - αΔx) e^{iθ_h}."""
    m = lat.bssh_chain(24, 9, walker=seed) if kind == "bssh" else lat.ossh_square(6, 12, walker=seed)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    g = np.random.default_rng(200 + seed)
    Nh = m.fpi.t.shape[0]
    theta = g.uniform(0, 2 * np.pi, Nh)
    fc = m.force_couplings(perm)
    ph = np.exp(1j * theta[perm[np.asarray(fc.s_bond) - 1] - 1])
    fc.s_alpha_im = fc.s_alpha * ph.imag
    fc.s_alpha2_im, fc.s_alpha3_im, fc.s_alpha4_im = np.zeros(len(ph)), np.zeros(len(ph)), np.zeros(len(ph))
    fc.s_alpha = fc.s_alpha * ph.real
    V0, _ = m.bare_model()
    return m, nt, perm, colors, fc, V0, np.exp(1j * theta)


@pytest.mark.parametrize("kind", ["bssh", "ossh"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_force_and_update_from_phonons_complex_T(kind, is_sym):
    """mul_νRe∂M∂x! (src/fermion_det_matrix_dervative.jl:2-245, generic in T) and the device-side update! from the phonon fields for
    T = ComplexF64, against the oracle (itself pinned by finite differences of the dense complex action,
    tests/test_oracle_complex_T.py::test_force_complex_T_against_finite_differences)."""
    m, nt, perm, colors, fc, V0, t0 = _flux_ssh(kind)
    Lt, N, Nh = m.fpi.Ltau, m.fpi.N, m.fpi.t.shape[0]
    h = L.Handle(Lt, N, nt, colors, is_sym, 1, 1, is_complex=True)
    cs, keep = L.couplings_struct(fc)
    h.call("smoqy_force_set_couplings", C.byref(cs))
    h.call("smoqy_set_bare_model", L.ptr(V0), L.ptr(np.ascontiguousarray(t0)), L.ptr(perm))
    x = np.asfortranarray(fc.x)
    h.call("smoqy_update_from_phonons_all", L.ptr(np.ascontiguousarray(x.T[None])))  # Nph x Ltau column-major == (Lt, Nph) C order per walker
    # fields: device update! from x against the oracle's V(x), t(x) -> exp / cosh / sinh
    V, t = orc.fields_from_phonons(fc, V0, t0, perm)
    expV, ch, sh = orc.update_fields(V, t, perm, m.fpi.dtau, is_sym)
    g_e = np.zeros((Lt, N), order="F")
    g_c, g_s = np.zeros((Lt, Nh), dtype=np.complex128, order="F"), np.zeros((Lt, Nh), dtype=np.complex128, order="F")
    h.call("smoqy_get_fields", 0, L.ptr(g_e), L.ptr(g_c), L.ptr(g_s))
    np.testing.assert_allclose(g_e, expV, rtol=1e-14)
    np.testing.assert_allclose(g_c.real, ch, rtol=1e-13)
    np.testing.assert_allclose(g_s, sh, rtol=1e-12, atol=1e-15)
    assert np.abs(g_s.imag).max() > 1e-3
    # force: ν Re<u|∂M/∂x|v> with random u, v
    o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    e = orc.OracleElph(fc)
    uv = rand(Lt, N, 2, 17)
    want = orc.mul_dMdx(o, e, colors, -2.0, uv[:, :, 0], uv[:, :, 1])
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, np.asfortranarray(uv[:, :, :1]))
    h.vec_upload(b, np.asfortranarray(uv[:, :, 1:]))
    out = np.zeros((1, Lt, fc.x.shape[0]))
    h.call("smoqy_force_dMdx_v", C.c_double(-2.0), a, b, L.ptr(out))
    assert np.abs(out[0].T - want).max() < 1e-11 * np.abs(want).max()
    # the whole force evaluation behind a solve (smoqy_pff_step_v) runs and gives a finite force on the free modes only
    phi, psi = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(phi, np.asfortranarray(uv[:, :, :1]))
    rv = np.ascontiguousarray(np.random.default_rng(5).standard_normal(2 * N))  # complex T: N complex deviates
    sf, it, ep = np.zeros(1), np.zeros(1, dtype=np.int32), np.zeros(1)
    dS = np.zeros((1, Lt, fc.x.shape[0]))
    h.call("smoqy_pff_step_v", phi, psi, None, L.ptr(rv), C.c_double(1e-10), 10000, 1, L.ptr(sf), L.ptr(it), L.ptr(ep), L.ptr(dS))
    Psi = np.asarray(h.vec_download(psi)).reshape(Lt, N, order="F")
    Lam = orc.update_lambda(Lt, N, x, m.elph.dtau, [], [], [], [], [])
    LP = orc.lambda_apply(Lam, Psi, "mul")
    AP = o.mul_M(LP)
    wantF = orc.mul_dMdx(o, e, colors, -2.0, AP, LP)
    assert ep[0] < 1e-10 and np.abs(dS[0].T - wantF).max() < 1e-9 * np.abs(wantF).max()
    if kind == "bssh":
        assert np.all(dS[0][:, -1] == 0)  # the infinite-mass partner mode receives no force
    h.close()


@pytest.mark.parametrize("kind", ["bssh", "ossh"])
def test_device_trajectory_complex_T(kind):
    """smoqy_hmc_trajectory_v on a complex handle (Nt steps in one call; the Lanczos start vectors are N COMPLEX deviates per walker and
    step, i.e. 2N doubles, sent in one transfer) against the same steps driven one by one through smoqy_pff_step_v + smoqy_efa_evolve on a
    second handle: the same solves on the same fields, hence equal iteration counts, actions, positions and momenta."""
    m, nt, perm, colors, fc, V0, t0 = _flux_ssh(kind)
    Lt, N = m.fpi.Ltau, m.fpi.N
    Nph = fc.x.shape[0]
    Nt, dt, nw = 3, 0.13, 1
    fm = np.asarray(fc.finite_mass, dtype=bool)
    row = m.fpi.dtau * (1.0 + 4.0 / m.fpi.dtau**2 * np.sin(np.pi * np.arange(Lt) / Lt) ** 2)
    q = np.asfortranarray(np.where(fm[:, None], row[None, :], np.inf))
    g = np.random.default_rng(11)
    Rphi = np.asfortranarray((g.standard_normal((Lt, N, 1)) + 1j * g.standard_normal((Lt, N, 1))) * np.sqrt(0.5))
    R = np.ascontiguousarray(g.standard_normal((nw, Lt, Nph)))
    rv = np.ascontiguousarray(g.standard_normal((Nt, nw, 2 * N)))
    hs = []
    for _ in range(2):
        h = L.Handle(Lt, N, nt, colors, True, nw, 1, is_complex=True)
        cs, keep = L.couplings_struct(fc)
        h.call("smoqy_force_set_couplings", C.byref(cs))
        h.call("smoqy_set_bare_model", L.ptr(V0), L.ptr(np.ascontiguousarray(t0)), L.ptr(perm))
        h.call("smoqy_update_from_phonons_all", L.ptr(np.ascontiguousarray(np.asfortranarray(fc.x).T[None])))
        h.call("smoqy_efa_config", L.ptr(q), L.ptr(q.copy(order="F")))
        phi, psi = h.vec_alloc(), h.vec_alloc()
        h.vec_upload(phi, Rphi)
        h.call("smoqy_matvec_v", L.OP_MT, phi, phi)
        h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, phi, phi)
        K = np.zeros(nw)
        h.call("smoqy_efa_initialize_momentum", L.ptr(R), L.ptr(K))
        hs.append((h, phi, psi, keep))
    (ha, pa, sa, _), (hb, pb, sb, _) = hs
    sf = np.zeros((Nt, nw)); it = np.zeros((Nt, nw), dtype=np.int32); ep = np.zeros((Nt, nw))
    ha.call("smoqy_hmc_trajectory_v", pa, sa, Nt, C.c_double(dt), C.c_double(1e-11), 10000, 1, L.ptr(rv), L.ptr(sf), L.ptr(it), L.ptr(ep))
    sf2 = np.zeros((Nt, nw)); it2 = np.zeros((Nt, nw), dtype=np.int32)
    hb.call("smoqy_efa_evolve", C.c_double(dt / 2), C.c_double(0.0), 1)
    for t in range(Nt):
        s1, i1, e1 = np.zeros(nw), np.zeros(nw, dtype=np.int32), np.zeros(nw)
        dS = np.zeros((nw, Lt, Nph))
        hb.call("smoqy_pff_step_v", pb, sb, None, L.ptr(np.ascontiguousarray(rv[t])), C.c_double(1e-11), 10000, 1, L.ptr(s1), L.ptr(i1), L.ptr(e1), L.ptr(dS))
        sf2[t], it2[t] = s1, i1
        hb.call("smoqy_efa_evolve", C.c_double(dt / 2 if t == Nt - 1 else dt), C.c_double(dt), 1)
    assert np.array_equal(it, it2) and it.min() > 0 and ep.max() < 1e-11
    np.testing.assert_allclose(sf, sf2, rtol=1e-10)
    xa, pa_ = np.zeros((nw, Lt, Nph)), np.zeros((nw, Lt, Nph))
    xb, pb_ = np.zeros((nw, Lt, Nph)), np.zeros((nw, Lt, Nph))
    ha.call("smoqy_efa_get_state", L.ptr(xa), L.ptr(pa_))
    hb.call("smoqy_efa_get_state", L.ptr(xb), L.ptr(pb_))
    np.testing.assert_allclose(xa, xb, atol=1e-11 * np.abs(xb).max())
    np.testing.assert_allclose(pa_, pb_, atol=1e-11 * np.abs(pb_).max())
    assert np.abs(xa - np.asfortranarray(fc.x).T[None]).max() > 1e-4  # the fields did move
    ha.close(); hb.close()
