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


# ---- four-point contractions (src/Measurements/GreensEstimator.jl:241-652) ---------------------------------

def _bconj(x, flag):
    """bconj (:729): conjugate when the flag is set."""
    return np.conj(x) if flag else x


def _shift(a, r):
    """ShiftedArrays.circshift(a, (0, (-r)..., [0])): result[i] = a[i + r] along the lattice axes 1..D."""
    return np.roll(a, shift=tuple(-int(x) for x in r), axis=tuple(range(1, 1 + len(r))))


def measure_CD0(C, AD, BD, C0, D0, tD=None, t0=None, conj_tD=False, conj_t0=False):
    """_measure_CΔ0! (:610-652): C[r] += (A[i+r]·B[i+r]) ⋆ (C[i]·D[i]) with optional hopping weights."""
    AB = AD * BD if tD is None else _bconj(tD, conj_tD) * AD * BD          # :626-635
    CD = C0 * D0 if t0 is None else _bconj(t0, conj_t0) * C0 * D0          # :637-646
    return translational_average(C, AB, CD)                                 # :649


def _pairs(Nrv):
    return [(n, m) for n in range(Nrv - 1) for m in range(n + 1, Nrv)]


def _views(GR, Rt, orbitals, rs):
    a, b, c, d = orbitals
    r1, r2, r3, r4 = rs
    return _shift(GR[:, a - 1], r1), _shift(Rt[:, b - 1], r2), _shift(GR[:, c - 1], r3), _shift(Rt[:, d - 1], r4)


def _mod1(x, L):
    return (int(x) - 1) % int(L)  # 0-based position of Julia's mod1(x, L)


def measure_GD0_GD0(GR, Rt, orbitals, r1, r2, r3, r4, tD=None, t0=None, conj_tD=False, conj_t0=False):
    """measure_GΔ0_GΔ0! (:241-388) up to add_contraction_to_correlation!; returns CΔ0 (Lτ+1, L...)."""
    Lt, Nrv = GR.shape[0], GR.shape[-1]
    Ls = GR.shape[2:-1]
    D = len(Ls)
    a, b, c, d = orbitals
    GRa, Rtb, GRc, Rtd = _views(GR, Rt, orbitals, (r1, r2, r3, r4))
    C = np.zeros((Lt + 1,) + Ls, dtype=complex)
    for n, m in _pairs(Nrv):                                                # :285-303
        measure_CD0(C, GRa[..., n], GRc[..., m], Rtb[..., n], Rtd[..., m], tD, t0, conj_tD, conj_t0)
    C /= len(_pairs(Nrv))                                                   # :306
    GR_a, Rt_b, GR_c, Rt_d = GR[:, a - 1], Rt[:, b - 1], GR[:, c - 1], Rt[:, d - 1]
    lat_axes = tuple(range(1, 1 + D))
    if a == b:                                                              # :312-337
        for i in range(Nrv):
            sh = np.roll(GR_c[..., i], shift=tuple(int(r1[k] - r2[k] - r3[k] + r4[k]) for k in range(D)), axis=lat_axes)
            idx = (Lt,) + tuple(_mod1(1 - r1[k] + r2[k], Ls[k]) for k in range(D))
            if tD is None and t0 is None:
                C[idx] -= np.sum(sh * Rt_d[..., i]) / (Nrv * sh.size)
            else:
                tb = np.roll(tD, shift=tuple(int(r1[k] - r2[k]) for k in range(D)), axis=lat_axes)
                C[idx] -= np.sum(_bconj(tb, conj_tD) * _bconj(t0, conj_t0) * sh * Rt_d[..., i]) / (Nrv * sh.size)
    if c == d:                                                              # :341-364
        for i in range(Nrv):
            sh = np.roll(GR_a[..., i], shift=tuple(int(-r1[k] + r2[k] + r3[k] - r4[k]) for k in range(D)), axis=lat_axes)
            idx = (Lt,) + tuple(_mod1(1 - r3[k] + r4[k], Ls[k]) for k in range(D))
            if tD is None and t0 is None:
                C[idx] -= np.sum(sh * Rt_b[..., i]) / (Nrv * sh.size)
            else:
                tb = np.roll(tD, shift=tuple(int(r3[k] - r4[k]) for k in range(D)), axis=lat_axes)
                C[idx] -= np.sum(_bconj(tb, conj_tD) * _bconj(t0, conj_t0) * sh * Rt_b[..., i]) / (Nrv * sh.size)
    if a == b and c == d and all((r2[k] - r1[k]) % Ls[k] == (r4[k] - r3[k]) % Ls[k] for k in range(D)):   # :367-382
        idx = (Lt,) + tuple(_mod1(1 + r2[k] - r1[k], Ls[k]) for k in range(D))
        if tD is None and t0 is None:
            C[idx] += 1
        else:
            tb = np.roll(tD, shift=tuple(int(r1[k] - r2[k]) for k in range(D)), axis=lat_axes)
            C[idx] += np.sum(_bconj(tb, conj_tD) * _bconj(t0, conj_t0)) / tb.size
    return C


def measure_GDD_G00(GR, Rt, orbitals, r1, r2, r3, r4, tD=None, t0=None, conj_tD=False, conj_t0=False):
    """measure_GΔΔ_G00! (:396-467)."""
    Lt, Nrv = GR.shape[0], GR.shape[-1]
    Ls = GR.shape[2:-1]
    GRa, Rtb, GRc, Rtd = _views(GR, Rt, orbitals, (r1, r2, r3, r4))
    C = np.zeros((Lt + 1,) + Ls, dtype=complex)
    for n, m in _pairs(Nrv):                                                # :439-457
        measure_CD0(C, GRa[..., n], Rtb[..., n], GRc[..., m], Rtd[..., m], tD, t0, conj_tD, conj_t0)
    C /= len(_pairs(Nrv))
    return C


def measure_G0D_GD0(GR, Rt, orbitals, r1, r2, r3, r4, tD=None, t0=None, conj_tD=False, conj_t0=False):
    """measure_G0Δ_GΔ0! (:475-606)."""
    Lt, Nrv = GR.shape[0], GR.shape[-1]
    Ls = GR.shape[2:-1]
    D = len(Ls)
    a, b, c, d = orbitals
    GRa, Rtb, GRc, Rtd = _views(GR, Rt, orbitals, (r1, r2, r3, r4))
    C = np.zeros((Lt + 1,) + Ls, dtype=complex)
    for n, m in _pairs(Nrv):                                                # :518-536
        measure_CD0(C, Rtb[..., n], GRc[..., m], GRa[..., n], Rtd[..., m], tD, t0, conj_tD, conj_t0)
    C /= len(_pairs(Nrv))
    GR_a, Rt_b, GR_c, Rt_d = GR[:, a - 1], Rt[:, b - 1], GR[:, c - 1], Rt[:, d - 1]
    lat_axes = tuple(range(1, 1 + D))
    if a == b:                                                              # :545-569, τ = 0
        for i in range(Nrv):
            sh = np.roll(GR_c[..., i], shift=tuple(int(-r1[k] + r2[k] - r3[k] + r4[k]) for k in range(D)), axis=lat_axes)
            idx = (0,) + tuple(_mod1(1 + r1[k] - r2[k], Ls[k]) for k in range(D))
            if tD is None and t0 is None:
                C[idx] -= np.sum(sh * Rt_d[..., i]) / (Nrv * sh.size)
            else:
                tb = np.roll(tD, shift=tuple(int(-r1[k] + r2[k]) for k in range(D)), axis=lat_axes)
                C[idx] -= np.sum(_bconj(tb, conj_tD) * _bconj(t0, conj_t0) * sh * Rt_d[..., i]) / (Nrv * sh.size)
    if c == d:                                                              # :575-599, τ = β
        for i in range(Nrv):
            sh = np.roll(GR_a[..., i], shift=tuple(int(-r1[k] + r2[k] - r3[k] + r4[k]) for k in range(D)), axis=lat_axes)
            idx = (Lt,) + tuple(_mod1(1 + r4[k] - r3[k], Ls[k]) for k in range(D))
            if tD is None and t0 is None:
                C[idx] -= np.sum(sh * Rt_b[..., i]) / (Nrv * sh.size)
            else:
                tb = np.roll(tD, shift=tuple(int(-r4[k] + r3[k]) for k in range(D)), axis=lat_axes)
                C[idx] -= np.sum(_bconj(tb, conj_tD) * _bconj(t0, conj_t0) * sh * Rt_b[..., i]) / (Nrv * sh.size)
    return C


def pair_correlation_direct(S0, S1, S2, S3, second, tD=None, t0=None):
    """Definition-level restatement of the pair sum for known-answer tests:
    C[r] = 1/Npairs Σ_{n<m} 1/n₁ Σ_i (S0·S1·tΔ)[i + r] (S2·S3·t0)[i], slot k taking random vector n or m (second[k]),
    with plain periodic index arithmetic instead of FFTs.  Slots have shape (Lτ, L..., Nrv)."""
    Nrv = S0.shape[-1]
    shape = S0.shape[:-1]
    out = np.zeros(shape, dtype=complex)
    pick = lambda S, k, n, m: S[..., m if second[k] else n]
    for n, m in _pairs(Nrv):
        X = pick(S0, 0, n, m) * pick(S1, 1, n, m) * (1 if tD is None else tD)
        Y = pick(S2, 2, n, m) * pick(S3, 3, n, m) * (1 if t0 is None else t0)
        for r in np.ndindex(*shape):
            out[r] += np.sum(np.roll(X, shift=tuple(-x for x in r), axis=tuple(range(len(shape)))) * Y)
    return out / (len(_pairs(Nrv)) * X.size)
