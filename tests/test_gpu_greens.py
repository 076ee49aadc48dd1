"""GreensEstimator on the device: the batched update_greens_estimator! and the
measure_GΔ0! contraction through the C ABI against the numpy oracle (oracle/greens.py, itself pinned against the
definition in tests/test_oracle_greens.py) and against dense solves."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import dense, greens, oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def geometry(kind):
    if kind == "honeycomb":
        return lat.holstein_honeycomb(3, 10), 2, (3, 3)
    if kind == "chain":
        return lat.bssh_chain(12, 9), 1, (12,)
    return lat.ossh_square(4, 6), 1, (4, 4)


@pytest.mark.parametrize("kind", ["honeycomb", "chain", "square"])
@pytest.mark.parametrize("precond", [False, True])
def test_update_and_measure_against_oracle(kind, precond):
    m, n, Ls = geometry(kind)
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-12)
    P = sq.KPMPreconditioner(fdm, rng=np.random.default_rng(2)) if precond else sq.I
    Nrv = 4
    ge = sq.GreensEstimator(fdm, (n, Ls), Nrv=Nrv, preconditioner=P, rng=np.random.default_rng(7), maxiter=5000, tol=1e-12)
    assert ge.GR.shape == (m.fpi.Ltau, n) + Ls + (Nrv,) and ge.Lτ == m.fpi.Ltau and ge.N == int(np.prod(Ls))
    # |R| = 1 and GR = M⁻¹ R  (src/Measurements/GreensEstimator.jl:141-168)
    assert np.abs(np.abs(ge.Rt) - 1).max() < 1e-15
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    V = ge.V
    R = np.conj(ge.Rt).reshape(V, Nrv, order="F")
    want = np.linalg.solve(M, R)
    assert np.abs(ge.GR.reshape(V, Nrv, order="F") - want).max() < 1e-9 * np.abs(want).max()
    # the contraction on the device against the oracle on the same GR, Rt
    for a in range(1, n + 1):
        for b in range(1, n + 1):
            corr = np.full(Ls + (ge.Lτ + 1,), 0.25 + 0j)
            sq.measure_GΔ0(corr, ge, (a, b))
            ref = greens.add_contraction_to_correlation(np.full(Ls + (ge.Lτ + 1,), 0.25 + 0j), greens.measure_GD0(ge.GR, ge.Rt, a, b), 1.0)
            assert np.abs(corr - ref).max() < 1e-13 * max(1.0, np.abs(ref).max()), (kind, a, b)


def test_update_follows_the_fields_and_warm_starts():
    """A second update after the fields moved picks up the new fields (device-to-device copy) and starts from the
    previous GR (ldiv!(GR′, …) uses GR′ as the initial guess, :159)."""
    m, n, Ls = geometry("honeycomb")
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-11)
    ge = sq.GreensEstimator(fdm, (n, Ls), Nrv=3, rng=np.random.default_rng(1), maxiter=5000, tol=1e-11)
    m.elph.x[...] *= 0.5
    m.refresh_from_x()
    sq.update(fdm, m.fpi)
    avg = sq.update_greens_estimator(ge, fdm, rng=np.random.default_rng(9), maxiter=5000, tol=1e-11)
    assert avg > 0
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    R = np.conj(ge.Rt).reshape(ge.V, 3, order="F")
    want = np.linalg.solve(M, R)
    assert np.abs(ge.GR.reshape(ge.V, 3, order="F") - want).max() < 1e-8 * np.abs(want).max()


def test_estimator_converges_to_the_exact_greens_function():
    """Statistical sanity: with many random vectors the estimate approaches the exact translational average."""
    m, n, Ls = lat.bssh_chain(8, 6), 1, (8,)
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-10)
    ge = sq.GreensEstimator(fdm, (n, Ls), Nrv=400, rng=np.random.default_rng(3), maxiter=5000, tol=1e-10)
    corr = np.zeros(Ls + (7,), dtype=complex)
    sq.measure_GΔ0(corr, ge, (1, 1))
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    M, _ = dense.dense_M(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    exact = np.moveaxis(greens.exact_GD0(np.linalg.inv(M), 6, 1, Ls, 1, 1), 0, -1)
    assert np.abs(corr - exact).max() < 0.1  # 1/sqrt(400 · 48) noise level ~ 0.01 per element, a loose 10σ bound
    assert abs(corr[0, 0].real - exact[0, 0].real) < 0.05


def test_ge_errors():
    m, n, Ls = geometry("chain")
    h = L.Handle(9, 12, *lat.checkerboard_decomposition(m.fpi.neighbor_table)[::2], True, 1, 2)
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_ge_measure_GD0", 0, 0, 1, 1, None)                       # not configured
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_ge_config", 1, 1, L.ptr(np.asarray([5], dtype=np.int64)))  # 1 x 5 != 12 sites
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_ge_config", 1, 3, L.ptr(np.asarray([2, 2, 3], dtype=np.int64)))  # D = 3 needs a 4-dimensional transform
    h.call("smoqy_ge_config", 1, 1, L.ptr(np.asarray([12], dtype=np.int64)))
    v = h.vec_alloc()
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_ge_measure_GD0", v, v, 2, 1, L.ptr(np.zeros((10, 12), dtype=complex)))  # orbital out of range


@pytest.mark.parametrize("kind", ["honeycomb", "chain"])
@pytest.mark.parametrize("weights", [False, True])
def test_four_point_estimators_against_oracle(kind, weights):
    """measure_GΔ0_GΔ0!, measure_GΔΔ_G00!, measure_G0Δ_GΔ0! (pair sums on the device, boundary terms on the host arrays)
    against the numpy restatement on the same GR, Rt — orbitals, displacements, weights and the δ-conditions varied."""
    m, n, Ls = geometry(kind)
    D = len(Ls)
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-10)
    ge = sq.GreensEstimator(fdm, (n, Ls), Nrv=5, rng=np.random.default_rng(4), maxiter=5000, tol=1e-10)
    g = np.random.default_rng(6)
    wshape = (ge.Lτ,) + Ls
    tD = (g.standard_normal(wshape) + 1j * g.standard_normal(wshape)) if weights else None
    t0 = (g.standard_normal(wshape) + 1j * g.standard_normal(wshape)) if weights else None
    z = (0,) * D
    cases = [((1, 1, 1, 1), z, z, z, z), ((1, n, n, 1), tuple([1] + [0] * (D - 1)), tuple([0] * (D - 1) + [2]), z, tuple([-1] * D)),
             ((n, n, 1, 1), tuple([2] * D), tuple([1] * D), tuple([1] * D), z)]
    fns = [(sq.measure_GΔ0_GΔ0, greens.measure_GD0_GD0), (sq.measure_GΔΔ_G00, greens.measure_GDD_G00), (sq.measure_G0Δ_GΔ0, greens.measure_G0D_GD0)]
    for orbitals, r1, r2, r3, r4 in cases:
        for dev_fn, ref_fn in fns:
            for cflags in ((False, False), (True, False)) if weights else ((False, False),):
                corr = np.full(Ls + (ge.Lτ + 1,), 0.5 + 0j)
                dev_fn(corr, ge, orbitals, r1, r2, r3, r4, 0.7, tD, t0, *cflags)
                ref = greens.add_contraction_to_correlation(np.full(Ls + (ge.Lτ + 1,), 0.5 + 0j), ref_fn(ge.GR, ge.Rt, orbitals, r1, r2, r3, r4, tD, t0, *cflags), 0.7)
                assert np.abs(corr - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (kind, dev_fn.__name__, orbitals, r1, r2, r3, r4, cflags)


def test_pair_estimator_needs_two_vectors():
    m, n, Ls = geometry("chain")
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-8)
    ge = sq.GreensEstimator(fdm, (n, Ls), Nrv=1, rng=np.random.default_rng(4), maxiter=5000, tol=1e-8)
    with pytest.raises(L.SmoqyError):
        sq.measure_GΔΔ_G00(np.zeros(Ls + (ge.Lτ + 1,), dtype=complex), ge, (1, 1, 1, 1), (0,), (0,), (0,), (0,), 1.0)


def test_walker_batch_measurement():
    """WalkerBatch.measure_greens: nwalkers x Nrv right-hand sides in one batched solve on a follower handle, G(Δ,0) per walker
    against the oracle contraction of the same GR, R."""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    b = WalkerBatch("holstein_honeycomb_L4_Ltau40", nwalkers=2)
    Nrv = 3
    G, iters = b.measure_greens(Nrv, orbitals=(1, 2), tol=1e-11)
    assert G.shape == (2, 41, 4, 4) and np.all(iters > 0)
    hg, _, (r, gr, mtr), R = b._ge
    GRd = hg.vec_download(gr)
    Lt, N = b.Lt, b.N
    for w, m in enumerate(b.models):
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, b.perm, m.fpi.dtau, True)
        M, _ = dense.dense_M(b.nt, expV, ch, sh, True)
        Rw = np.asarray(R[:, :, w * Nrv : (w + 1) * Nrv])
        want = np.linalg.solve(M, Rw.reshape(Lt * N, Nrv, order="F"))
        got = GRd[:, :, w * Nrv : (w + 1) * Nrv].reshape(Lt * N, Nrv, order="F")
        assert np.abs(got - want).max() < 1e-8 * np.abs(want).max()
        shape = (Lt, 2, 4, 4, Nrv)
        ref = greens.measure_GD0(got.reshape(shape, order="F"), np.conj(Rw).reshape(shape, order="F"), 1, 2)
        assert np.abs(G[w] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
    # a sweep with the measurement included counts 27 + Nrv solves per walker
    b2 = WalkerBatch("holstein_honeycomb_L4_Ltau40", nwalkers=2, measure_nrv=2)
    b2.sweep()
    assert b2.solves_per_sweep == 29 and b2.stats.solves == 2 * 29
