"""CPU restatement of the exact-Fourier-acceleration leapfrog the HMC update drives.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The arithmetic lives in SmoQyDQMC's ``ExactFourierAccelerator`` (``initialize_momentum!``, ``evolve_eom!``,
``kinetic_energy``), whose source is not under /root/reference; what is restated here is fixed by the reference's call sites
(src/EFAPFFHMCUpdater.jl:130-206, 244) and by the published algorithm (Cohen-Stead et al., Phys. Rev. E 105, 065302 (2022), §IV:
Fourier acceleration): the bosonic action of a phonon mode with mass M and frequency Ω,

    S_b = Δτ/2 Σ_l [ M Ω² x_l² + M (x_{l+1} − x_l)² / Δτ² ]        (periodic in l),

is diagonal in the τ-Fourier basis, S_b = ½ Σ_ω q_ω |x̃_ω|² with q_ω = Δτ M [Ω² + 4/Δτ² sin²(πω/Lτ)]; with fictitious momenta of
per-mode mass m_ω (K = ½ Σ_ω |p̃_ω|²/m_ω) the flow of K + S_b is a rotation of every (x̃_ω, p̃_ω) pair with frequency √(q_ω/m_ω), which
``evolve_eom!`` applies exactly; m = q makes every mode turn with unit frequency ("exact" acceleration).  Everything below is written
with dense unitary DFT matrices, sharing no code with the device kernel; tests/test_oracle_efa.py pins it against a matrix
exponential of the Hamiltonian flow, energy conservation and time reversal.

Arrays: x, p are (Nph, Ltau) like the reference's ``x``; q, m are (Nph, Ltau) indexed [phonon, ω].
"""
from __future__ import annotations

import numpy as np


def dft(Lt):
    """Unitary periodic transform F[ω, l] = exp(-2πi ω l / Lτ)/√Lτ."""
    k = np.arange(Lt)
    return np.exp(-2j * np.pi * np.outer(k, k) / Lt) / np.sqrt(Lt)


def harmonic_tables(Omega, M, dtau, Lt):
    """(q, m) of the exact acceleration, m = q = Δτ M [Ω² + 4/Δτ² sin²(πω/Lτ)]; infinite masses give infinite entries (frozen modes)."""
    Omega, M = np.atleast_1d(np.asarray(Omega, dtype=float)), np.atleast_1d(np.asarray(M, dtype=float))
    om = np.arange(Lt)
    with np.errstate(invalid="ignore"):
        q = dtau * M[:, None] * (Omega[:, None] ** 2 + 4.0 / dtau**2 * np.sin(np.pi * om / Lt)[None, :] ** 2)
    return np.asfortranarray(q), np.asfortranarray(q.copy())


def _live(m):
    return np.isfinite(m) & (m > 0)


def evolve_eom(x, p, dt, q, m, force=None, kick=0.0):
    """[p -= kick·force]; exact flow of K + S_b for time dt (evolve_eom!, src/EFAPFFHMCUpdater.jl:150, 202).  Returns new (x, p)."""
    x, p = np.array(x, dtype=float), np.array(p, dtype=float)
    if force is not None:
        p = p - kick * np.asarray(force)
    Lt = x.shape[1]
    F = dft(Lt)
    xt, pt = x @ F.T, p @ F.T  # row p: F x_p
    live = _live(m)
    mm = np.where(live, m, 1.0)
    w = np.sqrt(np.where(live, q / mm, 0.0))
    c, s = np.cos(w * dt), np.sin(w * dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = np.where(w > 0, s / (mm * w), dt / mm)
    xn = np.where(live, c * xt + f1 * pt, xt)
    pn = np.where(live, c * pt - mm * w * s * xt, pt)
    Fi = F.conj().T
    xo, po = (xn @ Fi.T).real, (pn @ Fi.T).real
    frozen = ~live.any(axis=1)  # infinite-mass modes are not touched at all
    xo[frozen], po[frozen] = x[frozen], p[frozen]
    return xo, po


def initialize_momentum(R, m):
    """p = F⁻¹ √m F R for unit normal deviates R (covariance F⁻¹ diag(m) F); returns (p, K) (initialize_momentum!, :142)."""
    R = np.asarray(R, dtype=float)
    F = dft(R.shape[1])
    live = _live(m)
    pt = np.where(live, np.sqrt(np.where(live, m, 0.0)), 0.0) * (R @ F.T)
    p = (pt @ F.conj()).real
    return p, kinetic_energy(p, m)


def kinetic_energy(p, m):
    """K = ½ Σ_ω |p̃_ω|²/m_ω (kinetic_energy(p, efa), :244)."""
    pt = np.asarray(p, dtype=float) @ dft(np.shape(p)[1]).T
    live = _live(m)
    return float(0.5 * np.sum(np.where(live, np.abs(pt) ** 2 / np.where(live, m, 1.0), 0.0)))


def bosonic_action(x, q, m=None):
    """S_b = ½ Σ_ω q_ω |x̃_ω|² over the live modes."""
    xt = np.asarray(x, dtype=float) @ dft(np.shape(x)[1]).T
    live = np.isfinite(q) if m is None else _live(m)
    return float(0.5 * np.sum(np.where(live, np.where(live, q, 0.0) * np.abs(xt) ** 2, 0.0)))


def bosonic_action_direct(x, Omega, M, dtau):
    """The same action from its τ-space definition (no transform): the known answer bosonic_action is checked against."""
    x = np.asarray(x, dtype=float)
    Omega, M = np.atleast_1d(np.asarray(Omega, dtype=float))[:, None], np.atleast_1d(np.asarray(M, dtype=float))[:, None]
    dx = np.roll(x, -1, axis=1) - x
    return float(np.sum(dtau / 2 * (M * Omega**2 * x**2 + M * dx**2 / dtau**2)))
