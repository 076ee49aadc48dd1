"""Dense-matrix builders for the known-answer tests.  TEST INFRASTRUCTURE ONLY.

Everything here is assembled with explicit numpy matrix products straight from the reference
docstring definitions (src/FermionDetMatrix.jl:5-17, 36, 129) — deliberately sharing no code
with ``smoqy_oracle.c`` so the two can check each other.
"""
from __future__ import annotations

import numpy as np


def bond_factor(N, i, j, c, s):
    """N x N matrix of one checkerboard factor [[c, s], [conj(s), c]] on sites (i, j), 0-based
    (src/checkerboard_matrix_multiply.jl:60-68)."""
    F = np.eye(N, dtype=np.result_type(c, s, np.float64))
    F[i, i] = c
    F[j, j] = c
    F[i, j] = s
    F[j, i] = np.conj(s)
    return F


def gamma(N, nt, c, s):
    """Γ = F_Nh ⋯ F_2 F_1: ``checkerboard_lmul!`` with ``transposed=false`` applies bond 1 first."""
    G = np.eye(N, dtype=np.result_type(np.asarray(s).dtype, np.float64))
    for h in range(nt.shape[1]):
        G = bond_factor(N, int(nt[0, h]) - 1, int(nt[1, h]) - 1, c[h], s[h]) @ G
    return G


def propagators(nt, expV, cosh, sinh, is_sym=True):
    """List of the Ltau propagators: Sym ``B_l = Γ_l D_l Γ_l^H`` (src/FermionDetMatrix.jl:36,
    409), Asym ``B_l = D_l Γ_l`` (:129, :451)."""
    Lt, N = expV.shape
    Bs = []
    for l in range(Lt):
        G = gamma(N, nt, cosh[l], sinh[l])
        D = np.diag(expV[l])
        Bs.append(G @ D @ G.conj().T if is_sym else D @ G)
    return Bs


def dense_M(nt, expV, cosh, sinh, is_sym=True):
    """The V x V fermion determinant matrix of src/FermionDetMatrix.jl:5-17 in the reference's
    vector layout (element (l, i) of an Ltau x N column-major array is index l + Ltau*i)."""
    Lt, N = expV.shape
    Bs = propagators(nt, expV, cosh, sinh, is_sym)
    M = np.zeros((Lt, N, Lt, N), dtype=Bs[0].dtype)
    for l in range(Lt):
        M[l, :, l, :] += np.eye(N)
        if l == 0:
            M[0, :, Lt - 1, :] += Bs[0]
        else:
            M[l, :, l - 1, :] -= Bs[l]
    # reference layout: vec index = l + Lt*i  <=>  array[l, i] raveled in Fortran order
    M = M.transpose(1, 0, 3, 2).reshape(N * Lt, N * Lt)  # index = i*Lt + l
    return M, Bs


def vec(a):
    """(Ltau, N) array -> reference flat vector (tau fastest)."""
    return np.asarray(a).ravel(order="F")


def unvec(v, Lt, N):
    return np.asarray(v).reshape((Lt, N), order="F")


def lambda_dense(Lam):
    """Dense Λ of src/holstein_shift_matrix.jl:47-71: (Λ v)[l] = Λ[l+1] v[l+1], wrap
    (Λ v)[Lτ] = Λ[1] v[1]."""
    Lt, N = Lam.shape
    A = np.zeros((Lt, N, Lt, N))
    for l in range(Lt):
        lp = (l + 1) % Lt
        for n in range(N):
            A[l, n, lp, n] = Lam[lp, n]
    return A.transpose(1, 0, 3, 2).reshape(N * Lt, N * Lt)


def ft_dense(Lt):
    """Dense unitary antiperiodic transform of src/FourierTransformer.jl:15, 46-47:
    U[w, l] = exp(-i pi l / Lt) / sqrt(Lt) * exp(-2 pi i w l / Lt)."""
    l = np.arange(Lt)
    w = np.arange(Lt)
    return np.exp(-2j * np.pi * np.outer(w, l) / Lt) * (np.exp(-1j * np.pi * l / Lt) / np.sqrt(Lt))[None, :]
