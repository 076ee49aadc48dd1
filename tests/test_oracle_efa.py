"""oracle/efa.py (the EFA leapfrog restatement, parity unpinned — SmoQyDQMC's source is absent) against independent known answers:
the τ-space definition of the bosonic action, a dense matrix exponential of the Hamiltonian flow, energy conservation, time reversal,
the quarter-period map of the exact acceleration, and the covariance of the sampled momenta."""
import numpy as np
import pytest
from scipy.linalg import expm

from oracle import efa


def setup(Nph=3, Lt=10, seed=0, exact=True):
    g = np.random.default_rng(seed)
    Omega, M, dtau = g.uniform(0.5, 1.5, Nph), g.uniform(0.5, 2.0, Nph), 0.05
    q, m = efa.harmonic_tables(Omega, M, dtau, Lt)
    if not exact:  # a general symmetric positive mass table
        bump = g.uniform(0.5, 2.0, (Nph, Lt))
        m = m * 0.5 * (bump + np.roll(bump[:, ::-1], 1, axis=1))
    x, p = g.standard_normal((Nph, Lt)), g.standard_normal((Nph, Lt))
    return Omega, M, dtau, q, m, x, p


def test_action_eigenvalues_match_the_tau_space_definition():
    Omega, M, dtau, q, m, x, p = setup()
    assert abs(efa.bosonic_action(x, q) - efa.bosonic_action_direct(x, Omega, M, dtau)) < 1e-11 * efa.bosonic_action(x, q)
    np.testing.assert_allclose(q, np.roll(q[:, ::-1], 1, axis=1), rtol=1e-14)  # ω ↔ −ω symmetric


@pytest.mark.parametrize("exact", [True, False])
def test_evolve_matches_the_matrix_exponential_of_the_flow(exact):
    Omega, M, dtau, q, m, x, p = setup(exact=exact)
    Lt = x.shape[1]
    F = efa.dft(Lt)
    dt = 0.37
    xn, pn = efa.evolve_eom(x, p, dt, q, m)
    for k in range(x.shape[0]):
        Q = (F.conj().T @ np.diag(q[k]) @ F).real       # S_b = ½ xᵀ Q x
        Minv = (F.conj().T @ np.diag(1 / m[k]) @ F).real  # K = ½ pᵀ M̃⁻¹ p
        G = np.block([[np.zeros((Lt, Lt)), Minv], [-Q, np.zeros((Lt, Lt))]])  # d/dt (x, p) = (M̃⁻¹ p, −Q x)
        z = expm(G * dt) @ np.concatenate([x[k], p[k]])
        np.testing.assert_allclose(np.concatenate([xn[k], pn[k]]), z, atol=1e-11 * np.abs(z).max())


@pytest.mark.parametrize("exact", [True, False])
def test_energy_conservation_time_reversal_and_composition(exact):
    Omega, M, dtau, q, m, x, p = setup(Nph=4, Lt=12, seed=3, exact=exact)
    H0 = efa.kinetic_energy(p, m) + efa.bosonic_action(x, q)
    x1, p1 = efa.evolve_eom(x, p, 0.9, q, m)
    assert abs(efa.kinetic_energy(p1, m) + efa.bosonic_action(x1, q) - H0) < 1e-11 * H0
    xb, pb = efa.evolve_eom(x1, -p1, 0.9, q, m)
    np.testing.assert_allclose(xb, x, atol=1e-12)
    np.testing.assert_allclose(-pb, p, atol=1e-11)
    x2, p2 = efa.evolve_eom(*efa.evolve_eom(x, p, 0.4, q, m), 0.5, q, m)
    np.testing.assert_allclose(x2, x1, atol=1e-12)
    np.testing.assert_allclose(p2, p1, atol=1e-11)


def test_exact_acceleration_quarter_period_and_kick():
    Omega, M, dtau, q, m, x, p = setup()
    F = efa.dft(x.shape[1])
    xq, pq = efa.evolve_eom(x, p, np.pi / 2, q, m)  # every mode turns by 90 degrees: x̃ -> p̃/m, p̃ -> −m x̃
    np.testing.assert_allclose(xq @ F.T, (p @ F.T) / m, atol=1e-12)
    np.testing.assert_allclose(pq @ F.T, -m * (x @ F.T), atol=1e-10)
    f = np.random.default_rng(5).standard_normal(x.shape)
    xa, pa = efa.evolve_eom(x, p, 0.3, q, m, force=f, kick=0.2)
    xb, pb = efa.evolve_eom(x, p - 0.2 * f, 0.3, q, m)
    np.testing.assert_array_equal(xa, xb)
    np.testing.assert_array_equal(pa, pb)


def test_infinite_mass_modes_are_frozen():
    Omega, M, dtau = np.array([1.0, 1.0]), np.array([1.0, np.inf]), 0.05
    q, m = efa.harmonic_tables(Omega, M, dtau, 8)
    g = np.random.default_rng(1)
    x, p = g.standard_normal((2, 8)), g.standard_normal((2, 8))
    x[1] = 0.0
    xn, pn = efa.evolve_eom(x, p, 0.5, q, m)
    np.testing.assert_array_equal(xn[1], x[1])
    np.testing.assert_array_equal(pn[1], p[1])
    pm, K = efa.initialize_momentum(g.standard_normal((2, 8)), m)
    assert np.all(pm[1] == 0.0) and K > 0


def test_momentum_covariance_and_kinetic_energy():
    Omega, M, dtau, q, m, x, p = setup(Nph=1, Lt=6)
    F = efa.dft(6)
    Mt = (F.conj().T @ np.diag(m[0]) @ F).real
    g = np.random.default_rng(7)
    R = g.standard_normal((20000, 1, 6))
    P = np.array([efa.initialize_momentum(r, m)[0][0] for r in R])
    cov = P.T @ P / len(P)
    np.testing.assert_allclose(cov, Mt, atol=0.05 * np.abs(Mt).max())
    r = R[0]
    pm, K = efa.initialize_momentum(r, m)
    assert abs(K - 0.5 * np.sum(r**2)) < 1e-12 * K          # Parseval: K = ½ |R|² at the moment of sampling
    assert abs(K - 0.5 * pm[0] @ np.linalg.solve(Mt, pm[0])) < 1e-12 * K
