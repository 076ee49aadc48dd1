"""The C oracle against the dense definition-level matrices on irregular graphs: random non-bipartite graphs whose colours are
not perfect matchings, negative hoppings, Sym and Asym — the same inputs tests/test_gpu_irregular.py feeds to the HIP path."""
import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from oracle import dense, oracle as orc

lat = sq.lattice


def random_graph(N, max_degree, nbonds, seed):
    g = np.random.default_rng(seed)
    deg = np.zeros(N, dtype=int)
    have, bonds = set(), []
    tries = 0
    while len(bonds) < nbonds and tries < 100000:
        tries += 1
        i, j = int(g.integers(N)), int(g.integers(N))
        if i == j or (min(i, j), max(i, j)) in have or deg[i] >= max_degree or deg[j] >= max_degree:
            continue
        have.add((min(i, j), max(i, j)))
        bonds.append((i + 1, j + 1))
        deg[i] += 1
        deg[j] += 1
    return np.asfortranarray(np.array(bonds, dtype=np.int64).T)


@pytest.mark.parametrize("N,deg,nb,seed", [(11, 3, 13, 1), (17, 4, 26, 2), (23, 2, 18, 3)])
@pytest.mark.parametrize("is_sym", [True, False])
def test_oracle_operators_on_irregular_graphs(N, deg, nb, seed, is_sym):
    Lt = 5
    nt_model = random_graph(N, deg, nb, seed)
    nt, perm, colors = lat.checkerboard_decomposition(nt_model)
    g = np.random.default_rng(10 + seed)
    V = np.asfortranarray(0.8 * g.standard_normal((N, Lt)))
    t = np.asfortranarray(1.0 + 0.3 * g.standard_normal((nt.shape[1], Lt)))
    t[0, :] *= -1.0
    expV, ch, sh = orc.update_fields(V, t, perm, 0.05, is_sym)
    # update!(fdm, fpi) from its definition (src/FermionDetMatrix.jl:217-231)
    dk = 0.025 if is_sym else 0.05
    tp = t[np.asarray(perm) - 1]
    np.testing.assert_allclose(expV, np.exp(-0.05 * V).T, rtol=1e-15)
    np.testing.assert_allclose(ch, np.cosh(dk * np.abs(tp)).T, rtol=1e-15)
    np.testing.assert_allclose(sh, (np.sign(tp) * np.sinh(dk * np.abs(tp))).T, rtol=1e-14)
    M, _ = dense.dense_M(nt, expV, ch, sh, is_sym)
    o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    v = np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))
    vv = dense.vec(v)
    for fn, want in ((o.mul_M, M @ vv), (o.mul_Mt, M.conj().T @ vv), (o.mul_MtM, M.conj().T @ (M @ vv)), (o.mul_MMt, M @ (M.conj().T @ vv))):
        assert np.abs(dense.vec(fn(v)) - want).max() < 1e-13 * np.abs(want).max()
    x, it, eps = o.cg_solve(v, tol=1e-12, maxiter=20000)
    want = np.linalg.solve(M.conj().T @ M, vv)
    assert np.abs(dense.vec(x) - want).max() < 1e-8 * np.abs(want).max() and it > 0
