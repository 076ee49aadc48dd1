"""Matrix-element type T = ComplexF64 (complex hoppings) in the CPU oracle, pinned against dense matrices assembled from the
reference definitions: the bond factor becomes [[c, s], [conj(s), c]] with s = sign(conj t)·sinh(Δτ′|t|)
(src/checkerboard_matrix_multiply.jl:60-68, src/FermionDetMatrix.jl:224-231); everything else is unchanged.
These run BEFORE the HIP path is trusted with complex hoppings (tests/test_gpu_complex_T.py)."""
import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from oracle import dense, oracle as orc

lat = sq.lattice


def complex_model(kind, is_sym, seed=0, tau_dependent=True):
    """A lattice of lattice.py with a random U(1) phase on every hopping (a static flux pattern), optionally with the modulus varying
    in τ as in the SSH models."""
    if kind == "honeycomb":
        m = lat.holstein_honeycomb(2, 6, walker=seed)
    elif kind == "square":
        m = lat.ossh_square(4, 5, walker=seed)
    else:
        m = lat.bssh_chain(6, 7, walker=seed)
    g = np.random.default_rng(100 + seed)
    Nh, Lt = m.fpi.t.shape
    phase = np.exp(1j * g.uniform(0, 2 * np.pi, Nh))[:, None]
    mod = m.fpi.t * (1.0 + (0.2 * g.standard_normal((Nh, Lt)) if tau_dependent else 0.0))
    t = np.asfortranarray(mod * phase)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, t, perm, m.fpi.dtau, is_sym)
    return m, t, nt, perm, colors, expV, ch, sh


def rand_vec(Lt, N, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))


@pytest.mark.parametrize("is_sym", [True, False])
def test_update_fields_complex_definition(is_sym):
    m, t, nt, perm, colors, expV, ch, sh = complex_model("square", is_sym)
    dt = m.fpi.dtau / 2 if is_sym else m.fpi.dtau
    tp = t[perm - 1, :].T  # (Lt, Nh) in checkerboard order
    np.testing.assert_allclose(ch, np.cosh(dt * np.abs(tp)), rtol=1e-15)
    np.testing.assert_allclose(sh, np.conj(tp) / np.abs(tp) * np.sinh(dt * np.abs(tp)), rtol=1e-14)
    assert np.iscomplexobj(sh) and np.abs(sh.imag).max() > 1e-3
    # each factor has unit determinant: c² - |s|² = 1
    np.testing.assert_allclose(ch**2 - np.abs(sh) ** 2, 1.0, rtol=1e-13)


@pytest.mark.parametrize("kind", ["honeycomb", "square", "chain"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_matvec_against_dense_complex_T(kind, is_sym):
    m, t, nt, perm, colors, expV, ch, sh = complex_model(kind, is_sym)
    Lt, N = expV.shape
    M, Bs = dense.dense_M(nt, expV, ch, sh, is_sym)
    assert np.iscomplexobj(M) and np.abs(M.imag).max() > 1e-3
    if is_sym:  # Γ D Γᴴ is Hermitian
        np.testing.assert_allclose(Bs[1], Bs[1].conj().T, atol=1e-14)
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    v = rand_vec(Lt, N, 1)
    for fn, A in ((f.mul_M, M), (f.mul_Mt, M.conj().T), (f.mul_MtM, M.conj().T @ M), (f.mul_MMt, M @ M.conj().T)):
        got = dense.vec(fn(v))
        want = A @ dense.vec(v)
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-14 * np.abs(want).max())
    u = rand_vec(Lt, N, 2)
    lhs, rhs = np.vdot(u, f.mul_M(v)), np.vdot(f.mul_Mt(u), v)
    assert abs(lhs - rhs) < 1e-12 * abs(lhs)


def test_checkerboard_complex_inverse_transpose_interval():
    m, t, nt, perm, colors, expV, ch, sh = complex_model("square", True)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, True)
    v = rand_vec(Lt, N, 4)
    for tr in (False, True):
        w = f.checkerboard(f.checkerboard(v, transposed=tr), transposed=tr, inverse=True)
        np.testing.assert_allclose(w, v, atol=1e-13)
    l = 2
    G = dense.gamma(N, nt, ch[l], sh[l])
    np.testing.assert_allclose(f.checkerboard(v)[l], G @ v[l], atol=1e-13)
    # `transposed` only reverses the bond order; every factor is Hermitian, so that is the conjugate transpose
    np.testing.assert_allclose(f.checkerboard(v, transposed=True)[l], G.conj().T @ v[l], atol=1e-13)
    np.testing.assert_allclose(f.checkerboard(v, inverse=True)[l], np.linalg.solve(G, v[l]), atol=1e-12)
    c0 = (int(colors[0, 1]) - 1, int(colors[1, 1]))
    Gc = dense.gamma(N, nt[:, c0[0] : c0[1]], ch[l, c0[0] : c0[1]], sh[l, c0[0] : c0[1]])
    np.testing.assert_allclose(f.checkerboard(v, interval=c0)[l], Gc @ v[l], atol=1e-13)


@pytest.mark.parametrize("is_sym", [True, False])
def test_cg_against_dense_solve_complex_T(is_sym):
    m, t, nt, perm, colors, expV, ch, sh = complex_model("honeycomb", is_sym)
    Lt, N = expV.shape
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    A = M.conj().T @ M
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    b = rand_vec(Lt, N, 8)
    x, iters, eps = f.cg_solve(b, tol=1e-13, maxiter=2000)
    want = np.linalg.solve(A, dense.vec(b))
    assert 0 < iters < 2000 and eps < 1e-13
    np.testing.assert_allclose(dense.vec(x), want, rtol=0, atol=1e-10 * np.abs(want).max())


@pytest.mark.parametrize("is_sym", [True, False])
def test_kpm_preconditioner_complex_T(is_sym):
    """τ-independent complex hoppings: P⁻¹ is the inverse of MᴴM up to the Chebyshev truncation; the Lanczos bounds bracket the
    spectrum (Sym) / singular values (Asym) of the dense complex B̄; the complex start vector is used as drawn."""
    m, t, nt, perm, colors, expV, ch, sh = complex_model("honeycomb", is_sym, tau_dependent=False)
    expV[:, :] = expV[:1, :]
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    assert f.is_complex
    P = orc.OracleKPM(f, a1=8.0, a2=8.0)
    g = np.random.default_rng(9)
    P.update((g.standard_normal(N) + 1j * g.standard_normal(N)) * np.sqrt(0.5))
    assert P.active
    emin, emax = P.bounds
    M, Bs = dense.dense_M(nt, expV, ch, sh, is_sym)
    sv = np.linalg.svd(Bs[0], compute_uv=False)
    assert emin < sv.min() and emax > sv.max()
    A = M.conj().T @ M
    v = rand_vec(Lt, N, 10)
    tol = 5e-6 if is_sym else 5e-2
    want = np.linalg.solve(A, dense.vec(v))
    np.testing.assert_allclose(dense.vec(P.apply(v)), want, atol=tol * np.abs(want).max())
    # Lanczos extremes are Ritz values of the dense Hermitian operator
    a, b = P.lanczos()
    lo, hi = orc.tridiag_extremes(a, b)
    H = Bs[0] if is_sym else Bs[0].conj().T @ Bs[0]
    ev = np.linalg.eigvalsh(H)
    assert ev[0] - 1e-12 <= lo and hi <= ev[-1] + 1e-12


@pytest.mark.parametrize("is_sym", [True, False])
def test_preconditioned_cg_complex_T(is_sym):
    m = lat.holstein_honeycomb(3, 24, smooth=True)
    g = np.random.default_rng(5)
    Nh, Lt = m.fpi.t.shape
    t = np.asfortranarray(m.fpi.t * np.exp(1j * g.uniform(0, 2 * np.pi, Nh))[:, None])
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    expV, ch, sh = orc.update_fields(m.fpi.V, t, perm, m.fpi.dtau, is_sym)
    Lt, N = expV.shape
    f = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    P = orc.OracleKPM(f)
    P.update((g.standard_normal(N) + 1j * g.standard_normal(N)) * np.sqrt(0.5))
    assert P.active
    b = rand_vec(Lt, N, 14)
    x0, it0, e0 = f.cg_solve(b, tol=1e-12, maxiter=5000)
    x1, it1, e1 = f.cg_solve(b, precond=P, tol=1e-12, maxiter=5000)
    assert e0 < 1e-12 and e1 < 1e-12 and it1 < it0
    np.testing.assert_allclose(x1, x0, atol=1e-9 * np.abs(x0).max())
    u, v = rand_vec(Lt, N, 15), rand_vec(Lt, N, 16)
    herm_tol = 1e-10 if is_sym else 5e-2
    assert abs(np.vdot(u, P.apply(v)) - np.vdot(P.apply(u), v)) < herm_tol * abs(np.vdot(u, P.apply(v)))
    assert np.vdot(u, P.apply(u)).real > 0


# ---- force terms and update! from the phonon fields with T = ComplexF64 (round 3) -------------------------------------------
# mul_νRe∂M∂x! is generic in T<:Number (src/fermion_det_matrix_dervative.jl:2-10): the bond factors are [[c, s], [conj(s), c]] and the SSH
# term is ν Re[conj(u_j) ΔτdK v_i + conj(u_i) conj(ΔτdK) v_j] with a COMPLEX ΔτdK (:225-227, ssh_parameters.α::Vector{T}).  Pinned here
# against central finite differences of the action computed with dense complex matrices, before the HIP path is trusted with it.
def flux_ssh_model(kind, seed=0):
    """An SSH model of lattice.py threaded by a static flux: hopping h carries the phase e^{iθ_h}, t_h = (1 - αΔx) e^{iθ_h}, i.e. a complex
    bare hopping t⁰ = e^{iθ} and complex couplings α e^{iθ}."""
    m = lat.bssh_chain(6, 5, walker=seed) if kind == "bssh" else lat.ossh_square(4, 4, walker=seed)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    g = np.random.default_rng(200 + seed)
    Nh = m.fpi.t.shape[0]
    theta = g.uniform(0, 2 * np.pi, Nh)  # model hopping order
    fc = m.force_couplings(perm)
    ph = np.exp(1j * theta[perm[np.asarray(fc.s_bond) - 1] - 1])  # coupling c acts on sorted bond s_bond[c] = model hopping perm[...]
    fc.s_alpha_im = fc.s_alpha * ph.imag
    fc.s_alpha2_im, fc.s_alpha3_im, fc.s_alpha4_im = np.zeros(len(ph)), np.zeros(len(ph)), np.zeros(len(ph))
    fc.s_alpha = fc.s_alpha * ph.real
    V0, _ = m.bare_model()
    t0 = np.exp(1j * theta)
    return m, nt, perm, colors, fc, V0, t0


def _action_and_fields(m, nt, perm, fc, V0, t0, is_sym, Phi):
    V, t = orc.fields_from_phonons(fc, V0, t0, perm)
    expV, ch, sh = orc.update_fields(V, t, perm, m.fpi.dtau, is_sym)
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    Lt, N = expV.shape
    Lam = orc.update_lambda(Lt, N, m.elph.x, m.elph.dtau, [], [], [], [], [])
    A = M @ dense.lambda_dense(Lam)
    psi = np.linalg.solve(A.conj().T @ A, dense.vec(Phi))
    return float(np.vdot(dense.vec(Phi), psi).real), psi, Lam, (expV, ch, sh), t


@pytest.mark.parametrize("kind", ["bssh", "ossh"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_force_complex_T_against_finite_differences(kind, is_sym):
    m, nt, perm, colors, fc, V0, t0 = flux_ssh_model(kind)
    Lt, N = m.fpi.Ltau, m.fpi.N
    g = np.random.default_rng(3)
    Phi = g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N))
    S, psi, Lam, (expV, ch, sh), t = _action_and_fields(m, nt, perm, fc, V0, t0, is_sym, Phi)
    assert np.iscomplexobj(t) and np.abs(t.imag).max() > 0.1 and np.iscomplexobj(sh)
    o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    e = orc.OracleElph(fc)
    Psi = dense.unvec(psi, Lt, N)
    LPsi = orc.lambda_apply(Lam, Psi, "mul")
    APsi = o.mul_M(LPsi)
    F = orc.mul_dMdx(o, e, colors, -2.0, APsi, LPsi)       # src/PFFCalculator.jl:146-150 (no Holstein coupling: ∂Λ/∂x = 0)
    x = fc.x
    h = 1e-5
    scale = np.abs(F).max()
    nfree = m.elph.x.shape[0]
    for (p, l) in [(0, 0), (1, 2), (nfree - 1, Lt - 1), (2, 1)]:
        x0 = x[p, l]
        x[p, l] = x0 + h
        Sp = _action_and_fields(m, nt, perm, fc, V0, t0, is_sym, Phi)[0]
        x[p, l] = x0 - h
        Sm = _action_and_fields(m, nt, perm, fc, V0, t0, is_sym, Phi)[0]
        x[p, l] = x0
        fd = (Sp - Sm) / (2 * h)
        assert abs(F[p, l] - fd) < 2e-6 * scale, (kind, is_sym, p, l, F[p, l], fd)
