"""Irregular lattices through every fast path: random non-bipartite graphs whose colours are NOT perfect matchings
(sites without a bond in a colour are padded with identity self-bonds on the device), odd sizes, 2-6 colours, a
coordination-6 triangular lattice.  Operators, the KPM preconditioner apply and the preconditioned CG against the CPU
oracle; small batches exercise the owner-computes kernels, a batch of 12 the LDS-resident ones."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
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


def triangular(Lx, Ly):
    idx = lambda x, y: (x % Lx) + Lx * (y % Ly) + 1
    b = []
    for y in range(Ly):
        for x in range(Lx):
            b += [(idx(x, y), idx(x + 1, y)), (idx(x, y), idx(x, y + 1)), (idx(x, y), idx(x + 1, y + 1))]
    return np.asfortranarray(np.array(b, dtype=np.int64).T)


def build(nt_model, N, Lt, seed, nwalkers=1, nrhs=1, is_sym=True):
    g = np.random.default_rng(seed)
    nt, perm, colors = lat.checkerboard_decomposition(nt_model)
    Nh = nt.shape[1]
    h = L.Handle(Lt, N, nt, colors, is_sym, nwalkers, nrhs)
    oracles = []
    for w in range(nwalkers):
        V = np.asfortranarray(0.8 * g.standard_normal((N, Lt)))
        t = np.asfortranarray(1.0 + 0.3 * g.standard_normal((Nh, Lt)))
        t[0, :] *= -1.0  # a negative hopping
        expV, ch, sh = orc.update_fields(V, t, perm, 0.05, is_sym)
        h.call("smoqy_update_from_path_integral", w, L.ptr(V), L.ptr(t), L.ptr(perm), C.c_double(0.05))
        oracles.append(orc.OracleFDM(nt, expV, ch, sh, is_sym))
    return h, oracles, colors.shape[1]


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


CASES = [
    ("random N=23 deg<=3", lambda: random_graph(23, 3, 28, 1), 23),
    ("random N=37 deg<=4", lambda: random_graph(37, 4, 60, 2), 37),
    ("random N=64 deg<=2", lambda: random_graph(64, 2, 50, 3), 64),
    ("triangular 5x4", lambda: triangular(5, 4), 20),
]


@pytest.mark.parametrize("name,make,N", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("is_sym", [True, False])
def test_irregular_operators_precond_and_cg(name, make, N, is_sym):
    Lt = 11
    nt_model = make()
    for nw, nrhs in ((1, 2), (3, 4)):  # 2 systems: owner-computes kernels; 12 systems: LDS-resident kernels
        h, oracles, ncol = build(nt_model, N, Lt, 5, nw, nrhs, is_sym)
        nsys = nw * nrhs
        g = np.random.default_rng(6)
        v = np.asfortranarray(g.standard_normal((Lt, N, nsys)) + 1j * g.standard_normal((Lt, N, nsys)))
        a, b = h.vec_alloc(), h.vec_alloc()
        h.vec_upload(a, v)
        for Tc in (1, 2):
            h.call("smoqy_set_tau_chunk", Tc)
            for op, fn in ((L.OP_M, "mul_M"), (L.OP_MT, "mul_Mt"), (L.OP_MTM, "mul_MtM"), (L.OP_MMT, "mul_MMt")):
                h.call("smoqy_matvec_v", op, b, a)
                got = h.vec_download(b)
                for s in range(nsys):
                    assert relerr(got[:, :, s], getattr(oracles[s // nrhs], fn)(v[:, :, s])) < 1e-13, (name, ncol, Tc, fn, s)
        # KPM preconditioner (Lanczos needs N > 20 steps: every case has N >= 20)
        Ps = []
        for w in range(nw):
            rv = np.random.default_rng(20 + w).standard_normal(N)
            P = orc.OracleKPM(oracles[w])
            P.update(rv)
            h.call("smoqy_precond_update", w, L.ptr(rv))
            Ps.append(P)
        out = np.zeros_like(v)
        h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, nsys)
        for s in range(nsys):
            assert relerr(out[:, :, s], Ps[s // nrhs].apply(v[:, :, s])) < 1e-10, (name, ncol, s)
        # preconditioned CG
        x = np.zeros_like(v)
        iters = np.zeros(nsys, dtype=np.int32)
        eps = np.zeros(nsys)
        h.call("smoqy_cg_solve", L.ptr(x), L.ptr(v), 1, 0, nsys, C.c_double(1e-10), 20000, 1, L.ptr(iters), L.ptr(eps))
        for s in (0, nsys - 1):
            o = oracles[s // nrhs]
            xo, ito, _ = o.cg_solve(v[:, :, s], precond=Ps[s // nrhs], tol=1e-10, maxiter=20000)
            assert abs(int(iters[s]) - ito) <= max(2, ito // 20), (name, iters[s], ito)
            assert relerr(o.mul_MtM(x[:, :, s]), v[:, :, s]) < 1e-8
        h.close()
