"""CPU restatement of the GreensEstimator contractions (src/Measurements/GreensEstimator.jl) in numpy.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()); the product never imports this.
Parity unpinned: the reference's tests hold no golden vectors for these functions and Julia is absent
(DESIGN.md §2).  What pins it: tests/test_oracle_greens.py — with Nrv = V random vectors that form a unit-modulus
orthogonal set (columns of a DFT matrix) the estimator is exact, and it must then equal the translational average
of the dense G = M⁻¹ taken straight from the definition, antiperiodic wrap and the τ = β slice included.
"""
from __future__ import annotations

import numpy as np


def random_phases(rng, shape):
    """randn!(rng, R); R ./= abs.(R)   (:141-142)."""
    R = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    return R / np.abs(R)


def aperiodic(a):
    """_aperiodic_copyto! (:656-671): (Lτ, ...) -> (2Lτ, ...), second half negated."""
    return np.concatenate([a, -a], axis=0)


def translational_average(S, a, b):
    """_translational_average! (:677-708): S[r] += (1/n) Σ_i a[i+r] b[i] through FFTs; S has one more entry along
    the first axis (τ = β takes the value of τ = 0)."""
    Lt = S.shape[0] - 1
    a = np.fft.fftn(a)      # mul!(a, pfft!, a)    :686
    b = np.fft.ifftn(b)     # mul!(b, pifft!, b)   :687
    a = a * b               # :692
    a = np.fft.ifftn(a)     # :695
    S[:Lt] += a[:Lt]        # :698-700
    S[Lt] += a[0]           # :703-705
    return S


def measure_GD0(GR, Rt, a, b):
    """measure_GΔ0! (:179-233) up to add_contraction_to_correlation!: returns CΔ0 of shape (Lτ+1, L...).
    GR, Rt have shape (Lτ, n, L..., Nrv); a, b are 1-based orbitals."""
    Lt, Nrv = GR.shape[0], GR.shape[-1]
    Ls = GR.shape[2:-1]
    G = np.zeros((Lt + 1,) + Ls, dtype=complex)
    for i in range(Nrv):
        A = aperiodic(GR[:, a - 1, ..., i])     # :213
        B = aperiodic(Rt[:, b - 1, ..., i])     # :214
        translational_average(G, A, B)          # :217
    G /= Nrv                                    # :219
    G[Lt] = -G[Lt]                              # :221-224
    if a == b:
        G[(Lt,) + (0,) * len(Ls)] += 1          # :225-227
    return G


def add_contraction_to_correlation(correlation, contraction, coef):
    """add_contraction_to_correlation! (:712-726): τ moves from the first to the last axis."""
    correlation += coef * np.moveaxis(contraction, 0, -1)
    return correlation


def exact_GD0(G, Lt, n, Ls, a, b):
    """The quantity measure_GΔ0! estimates, from the definition: G is the dense (Lτ·N)x(Lτ·N) matrix M⁻¹ in the
    (τ fastest, then site) ordering of the reference's vectors, site = orbital + n·cell.
    G(r, τ) = 1/(Lτ Nc) Σ_{τ', i} s(τ'+τ) G[(a, i+r, (τ'+τ) mod Lτ), (b, i, τ')],  s = -1 across the antiperiodic wrap,
    and G(r, β) = δ_ab δ(r) - G(r, 0)."""
    Nc = int(np.prod(Ls))
    N = n * Nc
    G4 = G.reshape(Lt, N, Lt, N, order="F")  # [τ, site, τ', site']
    out = np.zeros((Lt + 1,) + tuple(Ls), dtype=complex)
    cells = np.arange(Nc).reshape(Ls, order="F")
    for r in np.ndindex(*Ls):
        shifted = cells
        for d, rd in enumerate(r):
            shifted = np.roll(shifted, -rd, axis=d)  # shifted[i] = cell index of i + r
        src = cells.reshape(-1, order="F")
        dst = shifted.reshape(-1, order="F")
        for tau in range(Lt):
            acc = 0.0
            for tp in range(Lt):
                t2 = tp + tau
                sgn = 1.0 if t2 < Lt else -1.0
                acc += sgn * G4[t2 % Lt, (a - 1) + n * dst, tp, (b - 1) + n * src].sum()
            out[(tau,) + r] = acc / (Lt * Nc)
    out[Lt] = -out[0]
    if a == b:
        out[(Lt,) + (0,) * len(Ls)] += 1
    return out
