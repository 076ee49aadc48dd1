"""Known-answer tests that pin the CPU oracle (oracle/smoqy_oracle.c) against dense matrices
assembled from the reference docstring definitions.

The reference ships no golden vectors and cannot run in the build container ("parity
unpinned"); these tests are what stands between the oracle and a restatement error.
"""
import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from oracle import dense, oracle as orc

lat = sq.lattice


def small_model(kind, is_sym, seed=0):
    if kind == "honeycomb":
        m = lat.holstein_honeycomb(2, 6, walker=seed)
    elif kind == "square":
        m = lat.ossh_square(4, 5, walker=seed)
    else:
        m = lat.bssh_chain(6, 7, walker=seed)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    return m, nt, perm, colors, expV, ch, sh


def rand_vec(Lt, N, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))


KINDS = ["honeycomb", "square", "chain"]


@pytest.mark.parametrize("kind", KINDS)
def test_checkerboard_decomposition_is_proper(kind):
    m, nt, perm, colors, *_ = small_model(kind, True)
    raw = m.fpi.neighbor_table
    assert np.array_equal(nt, raw[:, perm - 1])
    assert colors[0, 0] == 1 and colors[1, -1] == nt.shape[1]
    for c in range(colors.shape[1]):
        sl = nt[:, colors[0, c] - 1 : colors[1, c]]
        sites = sl.ravel()
        assert len(set(sites.tolist())) == sites.size  # bonds of one colour are disjoint
        if c:
            assert colors[0, c] == colors[1, c - 1] + 1
    expected = {"honeycomb": 3, "square": 4, "chain": 2}[kind]
    assert colors.shape[1] == expected


@pytest.mark.parametrize("is_sym", [True, False])
def test_update_fields_matches_definition(is_sym):
    m, nt, perm, colors, expV, ch, sh = small_model("square", is_sym)
    dt = m.fpi.dtau * (0.5 if is_sym else 1.0)
    t = m.fpi.t[perm - 1, :]
    np.testing.assert_allclose(expV, np.exp(-m.fpi.dtau * m.fpi.V.T), rtol=1e-15)
    np.testing.assert_allclose(ch, np.cosh(dt * np.abs(t)).T, rtol=1e-15)
    np.testing.assert_allclose(sh, (np.sign(t) * np.sinh(dt * np.abs(t))).T, rtol=1e-15)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("is_sym", [True, False])
def test_matvec_against_dense(kind, is_sym):
    m, nt, perm, colors, expV, ch, sh = small_model(kind, is_sym)
    Lt, N = expV.shape
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    v = rand_vec(Lt, N, 1)
    for fn, A in ((f.mul_M, M), (f.mul_Mt, M.conj().T), (f.mul_MtM, M.conj().T @ M), (f.mul_MMt, M @ M.conj().T)):
        got = dense.vec(fn(v))
        want = A @ dense.vec(v)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-14 * np.abs(want).max())


@pytest.mark.parametrize("is_sym", [True, False])
def test_adjoint_identity(is_sym):
    m, nt, perm, colors, expV, ch, sh = small_model("honeycomb", is_sym)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    u, v = rand_vec(Lt, N, 2), rand_vec(Lt, N, 3)
    lhs = np.vdot(u, f.mul_M(v))
    rhs = np.vdot(f.mul_Mt(u), v)
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


def test_checkerboard_inverse_and_transpose():
    m, nt, perm, colors, expV, ch, sh = small_model("square", True)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, True)
    v = rand_vec(Lt, N, 4)
    for tr in (False, True):
        w = f.checkerboard(f.checkerboard(v, transposed=tr), transposed=tr, inverse=True)
        np.testing.assert_allclose(w, v, atol=1e-13)
    # dense Γ for one slice
    l = 2
    G = dense.gamma(N, nt, ch[l], sh[l])
    np.testing.assert_allclose(f.checkerboard(v)[l], G @ v[l], atol=1e-13)
    np.testing.assert_allclose(f.checkerboard(v, transposed=True)[l], G.T @ v[l], atol=1e-13)
    # interval form: one colour only
    c0 = (int(colors[0, 1]) - 1, int(colors[1, 1]))
    Gc = dense.gamma(N, nt[:, c0[0] : c0[1]], ch[l, c0[0] : c0[1]], sh[l, c0[0] : c0[1]])
    np.testing.assert_allclose(f.checkerboard(v, interval=c0)[l], Gc @ v[l], atol=1e-13)


def test_lambda_ops_against_dense():
    g = np.random.default_rng(5)
    Lt, N = 7, 5
    Lam = np.asfortranarray(-np.exp(0.3 * g.standard_normal((Lt, N))))
    Lam[0] *= -1
    A = dense.lambda_dense(Lam)
    v = rand_vec(Lt, N, 6)
    x = dense.vec(v)
    np.testing.assert_allclose(dense.vec(orc.lambda_apply(Lam, v, "mul")), A @ x, atol=1e-13)
    np.testing.assert_allclose(dense.vec(orc.lambda_apply(Lam, v, "mulT")), A.T @ x, atol=1e-13)
    np.testing.assert_allclose(dense.vec(orc.lambda_apply(Lam, v, "ldiv")), np.linalg.solve(A, x), atol=1e-12)
    np.testing.assert_allclose(dense.vec(orc.lambda_apply(Lam, v, "ldivT")), np.linalg.solve(A.T, x), atol=1e-12)


def test_update_lambda_definition():
    m = lat.holstein_honeycomb(2, 6)
    h = m.elph.holstein
    Lam = orc.update_lambda(6, 8, m.elph.x, m.elph.dtau, h.coupling_to_phonon, h.coupling_to_site, h.alpha, h.alpha3, h.ph_sym_form)
    sign = -np.ones((6, 8))
    sign[0] = 1
    want = sign * np.exp(0.5 * m.elph.dtau * (h.alpha[None, :] * m.elph.x.T))
    np.testing.assert_allclose(Lam, want, rtol=1e-15)


@pytest.mark.parametrize("Lt", [5, 8, 12, 40, 100, 128])
def test_fourier_transformer(Lt):
    N = 3
    U = dense.ft_dense(Lt)
    ft = orc.OracleFT(Lt, N)
    v = rand_vec(Lt, N, 7)
    w = ft.forward(v)
    np.testing.assert_allclose(w, U @ v, atol=1e-12)
    np.testing.assert_allclose(ft.inverse(w), v, atol=1e-12)
    np.testing.assert_allclose(U.conj().T @ U, np.eye(Lt), atol=1e-12)  # unitary
    # diagonalises the antiperiodic tau shift with phases phi_w = 2 pi (w + 1/2) / Lt  (KPMPreconditioner.jl:220)
    S = np.roll(np.eye(Lt), 1, axis=0)
    S[0, Lt - 1] = -1.0  # (S v)[l] = v[l-1], antiperiodic wrap
    Dg = U @ S @ U.conj().T
    phi = 2 * np.pi * (np.arange(Lt) + 0.5) / Lt
    np.testing.assert_allclose(Dg, np.diag(np.exp(-1j * phi)), atol=1e-12)


@pytest.mark.parametrize("is_sym", [True, False])
def test_cg_against_dense_solve(is_sym):
    m, nt, perm, colors, expV, ch, sh = small_model("honeycomb", is_sym)
    Lt, N = expV.shape
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    A = M.conj().T @ M
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    b = rand_vec(Lt, N, 8)
    x, iters, eps = f.cg_solve(b, tol=1e-13, maxiter=2000)
    want = np.linalg.solve(A, dense.vec(b))
    assert 0 < iters < 2000 and eps < 1e-13
    np.testing.assert_allclose(dense.vec(x), want, rtol=0, atol=1e-10 * np.abs(want).max())
    # warm start from the solution converges immediately (ConjugateGradient.jl:133)
    x2, it2, _ = f.cg_solve(b, x0=x, tol=1e-10)
    assert it2 == 0


def test_equal_time_greens_function():
    """Diagonal blocks of M^-1 are the equal-time DQMC Green's functions
    (I + B_l ... B_1 B_Lt ... B_{l+1})^-1  (SURVEY.md §4 item 6)."""
    m, nt, perm, colors, expV, ch, sh = small_model("honeycomb", True)
    Lt, N = expV.shape
    M, Bs = dense.dense_M(nt, expV, ch, sh, True)
    Minv = np.linalg.inv(M).reshape(N, Lt, N, Lt)
    for l in (0, 3, Lt - 1):
        P = np.eye(N)
        for k in list(range(l + 1, Lt)) + list(range(0, l + 1)):
            P = Bs[k] @ P
        G = np.linalg.inv(np.eye(N) + P)
        np.testing.assert_allclose(Minv[:, l, :, l], G, atol=1e-11)


@pytest.mark.parametrize("is_sym", [True, False])
def test_kpm_preconditioner_tau_independent_fields(is_sym):
    """For tau-independent fields P^-1 M†M ≈ I up to the Chebyshev truncation, and the KPM sum
    matches the exact scalar function through an eigendecomposition of B̄."""
    m = lat.holstein_honeycomb(2, 8)
    m.elph.x[...] = m.elph.x[:, :1]
    m.refresh_from_x()
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    P = orc.OracleKPM(f, a1=8.0, a2=8.0)  # generous orders: truncation error tiny
    P.update(np.random.default_rng(9).standard_normal(N))
    assert P.active
    emin, emax = P.bounds
    # bounds bracket the spectrum (singular values for Asym) of B̄
    _, Bs = dense.dense_M(nt, expV, ch, sh, is_sym)
    sv = np.linalg.svd(Bs[0], compute_uv=False)
    assert emin < sv.min() and emax > sv.max()
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    A = M.conj().T @ M
    v = rand_vec(Lt, N, 10)
    # Sym: exact up to Chebyshev truncation.  Asym: the reference applies conj(coefs) as a
    # polynomial in B̄ rather than B̄ᵀ (src/KPMPreconditioner.jl:523-530), which equals M̃⁻ᴴ only
    # for symmetric B̄; B̄ = D Γ is symmetric only up to O(Δτ²) commutators, hence the looser bound.
    tol = 5e-6 if is_sym else 5e-2
    got = dense.vec(P.apply(f.mul_MtM(v)))
    np.testing.assert_allclose(got, dense.vec(v), atol=tol * np.abs(v).max())
    # and P^-1 is (numerically) the exact inverse of A here
    got2 = dense.vec(P.apply(v))
    want2 = np.linalg.solve(A, dense.vec(v))
    np.testing.assert_allclose(got2, want2, atol=tol * np.abs(want2).max())


def test_kpm_coefficients_reproduce_function():
    m, nt, perm, colors, expV, ch, sh = small_model("honeycomb", True)
    f = orc.OracleFDM(nt, expV, ch, sh, True)
    P = orc.OracleKPM(f, a1=6.0, a2=6.0)
    P.update(np.random.default_rng(11).standard_normal(f.N))
    emin, emax = P.bounds
    Lt = f.Lt
    for slot in (0, len(P.order) - 1):
        c = P.coefs(slot).real
        phi = 2 * np.pi * (slot + 0.5) / Lt
        b = np.linspace(emin, emax, 7)
        xs = (b - 0.5 * (emax + emin)) / (0.5 * (emax - emin))
        approx = np.polynomial.chebyshev.chebval(xs, c)
        exact = 1.0 / (b * b - 2 * b * np.cos(phi) + 1)
        np.testing.assert_allclose(approx, exact, rtol=1e-4)


def test_kpm_order_formula_and_lanczos():
    m, nt, perm, colors, expV, ch, sh = small_model("chain", True)
    f = orc.OracleFDM(nt, expV, ch, sh, True)
    P = orc.OracleKPM(f)
    rv = np.random.default_rng(12).standard_normal(f.N)
    P.update(rv)
    emin, emax = P.bounds
    Lt = f.Lt
    phi = 2 * np.pi * (np.arange((Lt + 1) // 2) + 0.5) / Lt
    phi = np.where(phi > np.pi, 2 * np.pi - phi, phi)
    want = np.maximum(1, np.floor((emax - emin) * (2.0 / phi + 1.0))).astype(int)  # a1 doubled for Sym (:263)
    assert np.array_equal(P.order, want)
    # Lanczos tridiagonal extremes are Ritz values of the dense B̄
    d, c, s = P.bbar()
    Bbar = dense.gamma(f.N, nt, c, s) @ np.diag(d) @ dense.gamma(f.N, nt, c, s).T
    ev = np.linalg.eigvalsh(Bbar)
    a, b = P.lanczos()
    lo, hi = orc.tridiag_extremes(a, b)
    T = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
    tv = np.linalg.eigvalsh(T)
    assert abs(lo - tv[0]) < 1e-12 and abs(hi - tv[-1]) < 1e-12
    assert ev[0] - 1e-12 <= lo and hi <= ev[-1] + 1e-12
    np.testing.assert_allclose(P.bbar_mul(rv.astype(complex)), Bbar @ rv, atol=1e-13)


@pytest.mark.parametrize("is_sym", [True, False])
def test_preconditioned_cg_same_solution_fewer_iterations(is_sym):
    m = lat.holstein_honeycomb(3, 24, smooth=True)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    P = orc.OracleKPM(f)
    P.update(np.random.default_rng(13).standard_normal(N))
    assert P.active
    b = rand_vec(Lt, N, 14)
    x0, it0, e0 = f.cg_solve(b, tol=1e-12, maxiter=5000)
    x1, it1, e1 = f.cg_solve(b, precond=P, tol=1e-12, maxiter=5000)
    assert e0 < 1e-12 and e1 < 1e-12
    assert it1 < it0
    np.testing.assert_allclose(x1, x0, atol=1e-9 * np.abs(x0).max())
    # P^-1 is Hermitian positive definite on a probe (Asym: Hermitian only up to O(Δτ²))
    u, v = rand_vec(Lt, N, 15), rand_vec(Lt, N, 16)
    herm_tol = 1e-10 if is_sym else 5e-2
    assert abs(np.vdot(u, P.apply(v)) - np.vdot(P.apply(u), v)) < herm_tol * abs(np.vdot(u, P.apply(v)))
    assert np.vdot(u, P.apply(u)).real > 0


@pytest.mark.parametrize("is_sym", [True, False])
@pytest.mark.parametrize("kind", ["honeycomb", "chain_odd"])
def test_kpm_real_vector_method(kind, is_sym):
    """ldiv!(u′, P, u) for REAL vectors (src/KPMPreconditioner.jl:288-352 Sym, :417-485 Asym): half the frequencies are evaluated,
    the other half is their complex conjugate, the result is the real part of the back-transform.
    (i) For a real input the antiperiodic transform obeys v[Lτ-1-ω] = conj(v[ω]) and the per-frequency operator of ω and Lτ-1-ω
        is the same real matrix (Sym: real coefficients in real B̄; Asym: p_c(B̄)·p_c̄(B̄) = |p_c(B̄)|²), so the real method must
        equal the complex method applied to (u + 0i), whose imaginary part must vanish — checked, not assumed, for both types.
    (ii) Known answer: for τ-independent fields P⁻¹ is the exact inverse of MᵀM up to the Chebyshev truncation, also on real vectors.
    Odd Lτ exercises the middle frequency, which the reference conjugates onto itself (:334)."""
    if kind == "honeycomb":
        m = lat.holstein_honeycomb(2, 12)
    else:
        m = lat.bssh_chain(6, 9)  # Lτ = 9: odd
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    # τ-independent fields so that (ii) holds
    m.fpi.V[:, :] = m.fpi.V[:, :1]
    m.fpi.t[:, :] = m.fpi.t[:, :1]
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    P = orc.OracleKPM(f, a1=8.0, a2=8.0)
    P.update(np.random.default_rng(21).standard_normal(N))
    assert P.active
    u = np.asfortranarray(np.random.default_rng(22).standard_normal((Lt, N)))
    got = P.apply_real(u)
    full = P.apply(u.astype(complex))
    scale = np.abs(full).max()
    assert np.abs(full.imag).max() < 1e-13 * scale          # (i) the complex method keeps a real vector real
    np.testing.assert_allclose(got, full.real, atol=1e-13 * scale)
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    A = (M.conj().T @ M).real
    want = np.linalg.solve(A, dense.vec(u.astype(complex)).real)
    tol = 5e-6 if is_sym else 5e-2  # Asym: conj(coefs) as a polynomial in B̄ instead of B̄ᵀ, see the complex test above
    np.testing.assert_allclose(dense.vec(got.astype(complex)).real, want, atol=tol * np.abs(want).max())
    # inactive preconditioner: plain copy (:349 / :480)
    Q = orc.OracleKPM(f)
    np.testing.assert_array_equal(Q.apply_real(u), u)
