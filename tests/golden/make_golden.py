#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz:  python tests/golden/make_golden.py

WHAT THESE FIXTURES ARE.  The reference (SmoQyElPhQMC.jl) is pure Julia, cannot run in the build image and ships no golden
vectors, so nothing here was produced by it.  Each file holds seeded inputs and the outputs of THIS repository's
checkers on them:
  * "dense_*" arrays come from definition-level numpy code (oracle/dense.py, oracle/greens.py::exact_GD0: explicit N x N bond
    factors, dense M, numpy.linalg) that shares no code with the C oracle;
  * "oracle_*" arrays come from the C restatement (oracle/smoqy_oracle.c) where no dense form exists (Lanczos / KPM state,
    iteration counts, force terms — the latter pinned by finite differences in tests/test_oracle_force.py).
They pin regressions of the oracle and of the HIP path (tests/test_golden.py) on inputs that never change; they are not
evidence of parity with the Julia code — DESIGN.md §2 ("parity unpinned") still applies.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import smoqyelphqmc_amd as sq  # noqa: E402
from oracle import dense, greens, oracle as orc  # noqa: E402

lat = sq.lattice

CASES = {
    "holstein_honeycomb_L4_Ltau5": (lambda: lat.holstein_honeycomb(4, 5), 2, (4, 4)),
    "ossh_square_L6_Ltau4": (lambda: lat.ossh_square(6, 4), 1, (6, 6)),
    "bssh_chain_L22_Ltau4": (lambda: lat.bssh_chain(22, 4), 1, (22,)),
}


def make(name):
    build, n_orb, Ls = CASES[name]
    m = build()
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N = m.fpi.Ltau, m.fpi.N
    g = np.random.default_rng(20251004)
    v = np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))
    b = np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))
    rv = g.standard_normal(N)
    out = dict(neighbor_table=m.fpi.neighbor_table, sorted_table=nt, perm=perm, colors=colors, V=m.fpi.V, t=m.fpi.t, x=m.elph.x, dtau=np.float64(m.fpi.dtau), v=v, b=b, randvec=rv,
               n_orb=np.int64(n_orb), L=np.asarray(Ls, dtype=np.int64))
    for is_sym in (True, False):
        tag = "sym" if is_sym else "asym"
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
        M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
        vv = dense.vec(v)
        out[f"dense_{tag}_M_v"] = dense.unvec(M @ vv, Lt, N)
        out[f"dense_{tag}_Mt_v"] = dense.unvec(M.conj().T @ vv, Lt, N)
        out[f"dense_{tag}_MtM_v"] = dense.unvec(M.conj().T @ (M @ vv), Lt, N)
        out[f"dense_{tag}_MMt_v"] = dense.unvec(M @ (M.conj().T @ vv), Lt, N)
        out[f"dense_{tag}_solve_MtM_b"] = dense.unvec(np.linalg.solve(M.conj().T @ M, dense.vec(b)), Lt, N)
        o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
        _, it_plain, _ = o.cg_solve(b, tol=1e-10, maxiter=20000)
        out[f"oracle_{tag}_cg_iters_plain"] = np.int64(it_plain)
        if N > 20:  # the Lanczos bound estimate takes 20 steps
            P = orc.OracleKPM(o)
            P.update(rv)
            _, it_pre, _ = o.cg_solve(b, precond=P, tol=1e-10, maxiter=20000)
            out[f"oracle_{tag}_kpm_active"] = np.int64(P.active)
            out[f"oracle_{tag}_kpm_bounds"] = np.asarray(P.bounds)
            out[f"oracle_{tag}_kpm_order"] = np.asarray(P.order, dtype=np.int64)
            out[f"oracle_{tag}_kpm_apply_v"] = P.apply(v)
            out[f"oracle_{tag}_cg_iters_kpm"] = np.int64(it_pre)
        if is_sym:
            G = np.linalg.inv(M)
            out["dense_sym_GD0_11"] = greens.exact_GD0(G, Lt, n_orb, Ls, 1, 1)
            if n_orb > 1:
                out["dense_sym_GD0_12"] = greens.exact_GD0(G, Lt, n_orb, Ls, 1, 2)
            fc = m.force_couplings(perm)
            e = orc.OracleElph(fc)
            out["oracle_sym_dMdx_b_v"] = orc.mul_dMdx(o, e, colors, 1.0, b, v)
    hol = m.elph.holstein
    if hol is not None:
        Lam = orc.update_lambda(Lt, N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
        out["Lambda"] = Lam
        D = dense.lambda_dense(Lam)
        out["dense_lambda_mul_v"] = dense.unvec(D @ dense.vec(v), Lt, N)
        out["dense_lambda_ldivT_v"] = dense.unvec(np.linalg.solve(D.T, dense.vec(v)), Lt, N)
    out["dense_ft_forward_v"] = dense.ft_dense(Lt) @ v  # acts along tau for every site
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    return out


if __name__ == "__main__":
    for name in CASES:
        o = make(name)
        print(name, len(o), "arrays")
