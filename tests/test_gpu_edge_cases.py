"""Edge cases of the C ABI on the GPU: degenerate sizes, missing bonds, mixed active / inactive
preconditioners in one batch, iteration limits, and the generic (fallback) kernels on a
decomposition with more colours than the register-resident kernels hold."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make(nt_raw, Lt, N, is_sym=True, seed=0, nw=1, nrhs=1, vscale=1.0):
    g = np.random.default_rng(seed)
    nt, perm, colors = lat.checkerboard_decomposition(nt_raw)
    Nh = nt.shape[1]
    h = L.Handle(Lt, N, nt, colors, is_sym, nw, nrhs)
    oracles = []
    for w in range(nw):
        V = np.asfortranarray(vscale * g.standard_normal((N, Lt)))
        t = np.asfortranarray(1.0 + 0.3 * g.standard_normal((Nh, Lt)))
        expV, ch, sh = orc.update_fields(V, t, perm, 0.05, is_sym)
        h.call("smoqy_update_from_path_integral", w, L.ptr(V), L.ptr(t), L.ptr(perm), C.c_double(0.05))
        oracles.append(orc.OracleFDM(nt, expV, ch, sh, is_sym))
    return h, oracles, nt, colors


def rand(Lt, N, count, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N, count)) + 1j * g.standard_normal((Lt, N, count)))


def solve(h, b, tol, maxiter, pre):
    n = b.shape[2]
    x = np.zeros_like(b)
    it = np.zeros(n, dtype=np.int32)
    eps = np.zeros(n)
    h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, n, C.c_double(tol), maxiter, pre, L.ptr(it), L.ptr(eps))
    return x, it, eps


@pytest.mark.parametrize("Lt", [1, 2, 3])
@pytest.mark.parametrize("is_sym", [True, False])
def test_tiny_time_extent(Lt, is_sym):
    h, o, *_ = make(lat.chain_neighbor_table(4), Lt, 4, is_sym)
    v = rand(Lt, 4, 1, 1)
    for op, fn in ((L.OP_M, o[0].mul_M), (L.OP_MT, o[0].mul_Mt), (L.OP_MTM, o[0].mul_MtM), (L.OP_MMT, o[0].mul_MMt)):
        out = np.zeros_like(v)
        h.call("smoqy_matvec", op, L.ptr(out), L.ptr(v), 0, 1)
        assert relerr(out[:, :, 0], fn(v[:, :, 0])) < 1e-13
    x, it, eps = solve(h, v, 1e-12, 500, 0)
    xo, ito, _ = o[0].cg_solve(v[:, :, 0], tol=1e-12, maxiter=500)
    assert relerr(x[:, :, 0], xo) < 1e-10 and abs(int(it[0]) - ito) <= 1


def test_single_bond_and_no_bonds():
    h, o, *_ = make(np.array([[1], [2]], dtype=np.int64), 6, 2)
    v = rand(6, 2, 1, 2)
    out = np.zeros_like(v)
    h.call("smoqy_matvec", L.OP_MTM, L.ptr(out), L.ptr(v), 0, 1)
    assert relerr(out[:, :, 0], o[0].mul_MtM(v[:, :, 0])) < 1e-13
    # no hoppings at all: M is the atomic-limit operator, B_l = diag(exp(-ΔτV_l))
    hb, ob, *_ = make(np.zeros((2, 0), dtype=np.int64), 5, 3)
    w = rand(5, 3, 1, 3)
    hb.call("smoqy_matvec", L.OP_MTM, L.ptr(out := np.zeros_like(w)), L.ptr(w), 0, 1)
    assert relerr(out[:, :, 0], ob[0].mul_MtM(w[:, :, 0])) < 1e-13
    x, it, eps = solve(hb, w, 1e-12, 200, 0)
    assert relerr(ob[0].mul_MtM(x[:, :, 0]), w[:, :, 0]) < 1e-10


def test_many_colours_use_the_generic_kernels():
    """A star graph needs one colour per bond: 7 colours > the 4 (FermionDetMatrix) / 6 (KPM) that the
    register-resident kernels hold, so every launch goes through the generic path."""
    N = 8
    star = np.array([[1] * (N - 1), list(range(2, N + 1))], dtype=np.int64)
    h, o, nt, colors = make(star, 24, N, True, seed=4, vscale=0.3)
    assert colors.shape[1] == N - 1
    v = rand(24, N, 1, 5)
    out = np.zeros_like(v)
    h.call("smoqy_matvec", L.OP_MTM, L.ptr(out), L.ptr(v), 0, 1)
    assert relerr(out[:, :, 0], o[0].mul_MtM(v[:, :, 0])) < 1e-13
    rv = np.random.default_rng(6).standard_normal(N)
    P = orc.OracleKPM(o[0], n=6)
    P.update(rv)
    h.call("smoqy_precond_config", C.c_double(0.10), 6, C.c_double(1.0), C.c_double(1.0))
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    x, it, eps = solve(h, v, 1e-11, 2000, 1)
    xo, ito, _ = o[0].cg_solve(v[:, :, 0], precond=P if P.active else None, tol=1e-11, maxiter=2000)
    assert relerr(x[:, :, 0], xo) < 1e-9 and abs(int(it[0]) - ito) <= 2


def test_mixed_active_and_inactive_preconditioners():
    """Walker 1 gets on-site energies wild enough that the Lanczos bounds fail the sanity test
    (src/KPMPreconditioner.jl:573): its preconditioner must act as the identity while walker 0's works."""
    nt_raw = lat.honeycomb_neighbor_table(3)
    g = np.random.default_rng(7)
    nt, perm, colors = lat.checkerboard_decomposition(nt_raw)
    Lt, N, Nh = 16, 18, nt.shape[1]
    h = L.Handle(Lt, N, nt, colors, True, 2, 1)
    oracles, pre = [], []
    for w, vs in enumerate((1.0, 40.0)):
        V = np.asfortranarray(vs * g.standard_normal((N, Lt)))
        t = np.asfortranarray(np.ones((Nh, Lt)))
        expV, ch, sh = orc.update_fields(V, t, perm, 0.05, True)
        h.call("smoqy_update_from_path_integral", w, L.ptr(V), L.ptr(t), L.ptr(perm), C.c_double(0.05))
        o = orc.OracleFDM(nt, expV, ch, sh, True)
        rv = g.standard_normal(N)
        P = orc.OracleKPM(o)
        P.update(rv)
        h.call("smoqy_precond_update", w, L.ptr(rv))
        oracles.append(o)
        pre.append(P)
    assert pre[0].active and not pre[1].active
    act = [C.c_int(0), C.c_int(0)]
    for w in range(2):
        h.call("smoqy_precond_get", w, C.byref(act[w]), None, None, None, None, None)
    assert act[0].value == 1 and act[1].value == 0
    v = rand(Lt, N, 2, 8)
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, 2)
    assert relerr(out[:, :, 0], pre[0].apply(v[:, :, 0])) < 1e-11
    assert relerr(out[:, :, 1], v[:, :, 1]) < 1e-13          # identity
    x, it, eps = solve(h, v, 1e-9, 20000, 1)
    for w in range(2):
        xo, ito, _ = oracles[w].cg_solve(v[:, :, w], precond=pre[w] if pre[w].active else None, tol=1e-9, maxiter=20000)
        assert abs(int(it[w]) - ito) <= max(3, ito // 10)  # ~1e4 iterations on the ill-conditioned walker: rounding-order sensitive
        # walker 1 is deliberately ill-conditioned (|x| ~ 1e7): the true residual drifts from the recursive one
        assert relerr(oracles[w].mul_MtM(x[:, :, w]), v[:, :, w]) < (1e-8 if w == 0 else 1e-4)


def test_iteration_limits_and_zero_rhs():
    h, o, *_ = make(lat.honeycomb_neighbor_table(2), 8, 8)
    b = rand(8, 8, 1, 9)
    x, it, eps = solve(h, b, 1e-12, 0, 0)      # maxiter = 0: nothing done, (0 == maxiter, eps0)
    assert it[0] == 0 and abs(eps[0] - 1.0) < 1e-14 and np.all(x == 0)
    x, it, eps = solve(h, b, 1e-30, 7, 0)      # unreachable tolerance: (maxiter, eps) without an error
    assert it[0] == 7 and eps[0] > 0
    x, it, eps = solve(h, np.zeros_like(b), 1e-10, 50, 0)  # b = 0 -> x = 0
    assert it[0] == 0 and np.all(x == 0)


def test_multi_rhs_with_preconditioner():
    m = lat.holstein_honeycomb(4, 40)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    h = L.Handle(40, 32, nt, colors, True, 1, 5)
    h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
    o = orc.OracleFDM(nt, expV, ch, sh, True)
    rv = np.random.default_rng(10).standard_normal(32)
    P = orc.OracleKPM(o)
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    b = rand(40, 32, 5, 11)
    b[:, :, 3] *= 1e-6                     # very different scales / convergence times in one batch
    x, it, eps = solve(h, b, 1e-10, 10000, 1)
    for s in range(5):
        xo, ito, _ = o.cg_solve(b[:, :, s], precond=P, tol=1e-10, maxiter=10000)
        assert abs(int(it[s]) - ito) <= 2 and relerr(x[:, :, s], xo) < 1e-8
    # sub-range host call: only systems 1..2 are touched
    x2 = np.zeros((40, 32, 2), dtype=complex, order="F")
    it2 = np.zeros(2, dtype=np.int32)
    e2 = np.zeros(2)
    h.call("smoqy_cg_solve", L.ptr(x2), L.ptr(np.asfortranarray(b[:, :, 1:3])), 1, 1, 2, C.c_double(1e-10), 10000, 1, L.ptr(it2), L.ptr(e2))
    assert relerr(x2, x[:, :, 1:3]) < 1e-9


def test_bitwise_reproducible_solves_and_force():
    """No atomics anywhere on the path: every reduction is a fixed-order tree over fixed-order partials, so two runs on the
    same inputs agree bit for bit — the solve (iterates, iteration counts, residuals) and the force."""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    outs = []
    for _ in range(2):
        b = WalkerBatch("holstein_honeycomb_L4_Ltau40", nwalkers=3)
        b.sample_pseudofermion_fields()
        sf, iters, eps = b.calculate_fermionic_action(1e-10)
        outs.append((b.h.vec_download(b.u).copy(), sf.copy(), iters.copy(), eps.copy(), b.fermionic_force().copy()))
        b.h.close()
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("Lt", [13, 22, 34])
@pytest.mark.parametrize("is_sym", [True, False])
def test_time_extents_with_large_prime_factors_fall_back_to_rocfft(Lt, is_sym):
    """Lτ with a prime factor above 7 has no Stockham plan: the preconditioner and the FourierTransformer go through the rocFFT
    plans and the unfused CG kernels automatically."""
    N = 24
    h, o, nt, colors = make(lat.chain_neighbor_table(N), Lt, N, is_sym, seed=3, nrhs=2, vscale=0.5)
    v = rand(Lt, N, 2, 4)
    w = v.copy(order="F")
    h.call("smoqy_fft_forward", L.ptr(w), 0, 2)
    ft = orc.OracleFT(Lt, N)
    assert relerr(w[:, :, 1], ft.forward(v[:, :, 1])) < 1e-13
    h.call("smoqy_fft_inverse", L.ptr(w), 0, 2)
    assert relerr(w, v) < 1e-13
    rv = np.random.default_rng(5).standard_normal(N)
    P = orc.OracleKPM(o[0])
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, 2)
    assert relerr(out[:, :, 0], P.apply(v[:, :, 0])) < 1e-10
    x, it, eps = solve(h, v, 1e-10, 5000, 1)
    xo, ito, _ = o[0].cg_solve(v[:, :, 0], precond=P, tol=1e-10, maxiter=5000)
    assert abs(int(it[0]) - ito) <= 2 and eps.max() < 1e-10
    assert relerr(x[:, :, 0], xo) < 1e-8


@pytest.mark.parametrize("use_graph", [0, 1])
def test_iterate_after_every_iteration_count(use_graph):
    """The x update of the fused CG iteration lives one kernel later than the r update (the inverse τ-FFT kernel applies x += αp): cut
    the solve off after 1 … 7 iterations (unreachable tolerance) and compare the iterate with the oracle's after the same number of
    iterations — eager launches and the four-iteration captured graph (whose replays run past maxiter as early-exit workgroups)."""
    N, Lt_ = 24, 12
    h, o, nt, colors = make(lat.chain_neighbor_table(N), Lt_, N, True, seed=13, nrhs=2, vscale=0.5)
    v = rand(Lt_, N, 2, 14)
    rv = np.random.default_rng(15).standard_normal(N)
    P = orc.OracleKPM(o[0])
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    h.call("smoqy_cg_use_graph", use_graph)
    for maxiter in range(1, 8):
        x, it, eps = solve(h, v, 1e-30, maxiter, 1)
        assert list(it) == [maxiter, maxiter]
        for s_ in range(2):
            xo, ito, _ = o[0].cg_solve(v[:, :, s_], precond=P, tol=1e-30, maxiter=maxiter)
            assert ito == maxiter
            assert relerr(x[:, :, s_], xo) < 1e-11, (maxiter, s_)


def _oversubscribed_solve(concurrent):
    """16 walkers x 10 right-hand sides at L = 16, Lτ = 128: 160 systems x 64 site tiles = 10 240 workgroups per τ-FFT launch, far
    beyond what is co-resident (256 CUs), optionally with a long-running kernel queue on a second stream competing for the CUs."""
    import threading

    name = "holstein_honeycomb_L16_Ltau128"
    nw, nrhs = 16, 10
    models = [lat.CONFIGS[name](walker=w) for w in range(nw)]
    nt, perm, colors = lat.checkerboard_decomposition(models[0].fpi.neighbor_table)
    Lt, N = models[0].fpi.Ltau, models[0].fpi.N
    h = L.Handle(Lt, N, nt, colors, True, nw, nrhs)
    for w, m in enumerate(models):
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    rv = np.ascontiguousarray(np.random.default_rng(77).standard_normal((nw, N)))
    h.call("smoqy_precond_update_all", L.ptr(rv))
    g = np.random.default_rng(78)
    ph = g.uniform(0, 2 * np.pi, (Lt, N, nw * nrhs))
    b = np.asfortranarray(np.exp(1j * ph))  # unit-modulus random-phase vectors (src/Measurements/GreensEstimator.jl:141-142)
    b[:, :, 5::7] *= 1e-3  # spread the convergence times so systems retire in different iterations
    bid, xid, aid = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(bid, b)
    h.vec_upload(xid, b)
    other = None
    if concurrent:
        h2 = L.Handle(Lt, N, nt, colors, True, 64, 1)
        for w in range(64):
            m = models[w % nw]
            h2.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
        u, v = h2.vec_alloc(), h2.vec_alloc()
        other = threading.Thread(target=lambda: h2.bench_matvec(L.OP_MTM, v, u, 4000))  # ~0.2 s of back-to-back full-chip launches
        other.start()
    it = np.zeros(nw * nrhs, dtype=np.int32)
    eps = np.zeros(nw * nrhs)
    h.call("smoqy_cg_solve_v", xid, xid, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
    if other:
        other.join()
        h2.close()
    h.call("smoqy_matvec_v", L.OP_MTM, aid, xid)
    x, ax = h.vec_download(xid), h.vec_download(aid)
    res = np.array([np.linalg.norm(ax[:, :, s] - b[:, :, s]) / np.linalg.norm(b[:, :, s]) for s in range(nw * nrhs)])
    h.close()
    return x, it, eps, res


def test_cg_stop_latch_on_an_oversubscribed_grid():
    """CgState::stop (kernels_tfft.hip / kernels_vec.hip): the closing kernel of a CG iteration must not gate on the `done` flag one
    of its own workgroups writes.  With 10 240 workgroups per launch a late workgroup of a system certainly starts after that
    system's tile 0 has finished; had it skipped the final x += αp the true residual would exceed the reported one by up to κ.
    Checks: true residual ‖b − MᵀMx‖/‖b‖ of every system agrees with the returned ϵ; two runs — one alone, one with a second
    stream saturating the chip — agree bit for bit."""
    x0, it0, eps0, res0 = _oversubscribed_solve(False)
    assert np.all(eps0 < 1e-10) and len(set(it0.tolist())) > 1  # systems did retire at different iterations
    assert np.all(res0 < 1e-10)
    np.testing.assert_allclose(res0, eps0, rtol=0.01, atol=2e-13)
    x1, it1, eps1, res1 = _oversubscribed_solve(True)
    assert np.array_equal(it0, it1) and np.array_equal(eps0, eps1)
    assert np.array_equal(x0, x1)


def test_graph_replay_survives_a_coefficient_table_reallocation():
    """A captured CG iteration bakes KpmArgs (d_coefs, maxorder) and the FFT choice into its kernel arguments; growing the table
    (order > maxorder through smoqy_precond_set) or flipping smoqy_fft_use_rocfft must drop the cached graphs instead of replaying
    them with a freed pointer."""
    m = lat.holstein_honeycomb(4, 40)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N = 40, 32

    def mk():
        h = L.Handle(Lt, N, nt, colors, True, 1, 2)
        h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
        h.call("smoqy_precond_update", 0, L.ptr(np.random.default_rng(10).standard_normal(N)))
        return h

    def state(h):
        act, norder = C.c_int(0), C.c_int(0)
        bounds, order = np.zeros(2), np.zeros(Lt, dtype=np.int32)
        h.call("smoqy_precond_get", 0, C.byref(act), bounds.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(norder), None, None)
        return bounds, order[: norder.value].copy()

    b = rand(Lt, N, 2, 12)
    outs = {}
    for graph in (0, 1):
        h = mk()
        h.call("smoqy_cg_use_graph", graph)
        x_a, it_a, _ = solve(h, b, 1e-10, 10000, 1)  # captures with the initial table (maxorder = 64)
        bounds, order = state(h)
        # a host-supplied expansion whose slot-0 order exceeds the table's stride: constant polynomials (coefficient 1, rest 0)
        big = order.copy()
        big[0] = 150
        coefs = np.zeros(int(big.sum()), dtype=complex)
        coefs[np.concatenate(([0], np.cumsum(big)[:-1]))] = 1.0
        h.call("smoqy_precond_set", 0, 1, L.ptr(bounds), L.ptr(big), L.ptr(coefs))
        x_b, it_b, eps_b = solve(h, b, 1e-10, 10000, 1)
        h.call("smoqy_fft_use_rocfft", 1)
        x_c, it_c, eps_c = solve(h, b, 1e-10, 10000, 1)
        en, cap = C.c_int(-1), C.c_int(-1)
        h.call("smoqy_cg_graph_status", C.byref(en), C.byref(cap))
        assert en.value == graph and (cap.value >= 1) == bool(graph)
        outs[graph] = (x_a, it_a, x_b, it_b, x_c, it_c)
        assert eps_b.max() < 1e-10 and eps_c.max() < 1e-10
        h.close()
    for eager, replay in zip(outs[0], outs[1]):
        assert np.array_equal(eager, replay)


def test_process_wide_cg_gate():
    """smoqy_cg_gate: with the limit at 1, two handles solved from two host threads take turns inside the CG loop and return exactly what
    they return alone; limit 0 switches the gate off again."""
    import threading

    lib = L.load()
    hs, bs, want = [], [], []
    for k in range(2):
        h, o, *_ = make(lat.chain_neighbor_table(24), 16, 24, True, seed=20 + k, nrhs=2, vscale=0.5)
        b = rand(16, 24, 2, 30 + k)
        hs.append(h); bs.append(b)
        want.append(solve(h, b, 1e-11, 5000, 0))
    try:
        assert lib.smoqy_cg_gate(1) == 0
        got = [None, None]

        def run(k):
            for _ in range(5):
                got[k] = solve(hs[k], bs[k], 1e-11, 5000, 0)

        ts = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts)
        for k in range(2):
            for a, b in zip(got[k], want[k]):
                assert np.array_equal(a, b)
    finally:
        lib.smoqy_cg_gate(0)


@pytest.mark.parametrize("nw,bitwise", [(6, True), (9, False), (20, False)])
def test_cg_pipeline_parts_agree(nw, bitwise):
    """smoqy_cg_split: the iteration kernels of the parts of a batch run on separate streams of the handle; per system the arithmetic is the
    same kernels on a sub-range.  When every part selects the same kernel family as the whole batch (6 systems: owner-computes either way) the
    results are bit-identical; when a part of <= 8 systems drops to the owner-computes MᵀM kernel while the whole batch uses the LDS-resident
    one (9 or 20 systems split in two / three) they agree to rounding with the same iteration counts."""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    outs = {}
    for parts in (1, 2, 3):
        b = WalkerBatch("holstein_honeycomb_L4_Ltau40", nwalkers=nw, cg_split=parts)
        b.sample_pseudofermion_fields()
        sf, iters, eps = b.calculate_fermionic_action(1e-10)
        outs[parts] = (b.h.vec_download(b.u).copy(), sf.copy(), iters.copy(), eps.copy(), b.fermionic_force().copy())
        b.h.close()
    for parts in (2, 3):
        assert np.array_equal(outs[1][2], outs[parts][2])  # iteration counts
        for x, y in zip(outs[1], outs[parts]):
            if bitwise:
                assert np.array_equal(x, y)
            else:
                np.testing.assert_allclose(x, y, rtol=1e-9, atol=1e-11 * np.abs(x).max())


@pytest.mark.parametrize("entry", ["path_integral", "fields"])
def test_tau_independent_hoppings_select_and_leave_the_one_pair_kernel(entry):
    """The Sym LDS-resident MᵀM kernel keeps one (cosh, sinh) pair per colour once the HOST has seen that no walker of the launch has
    τ-dependent hoppings (FermionDetMatrix.jl:224-231 computes them per slice; api_handle.hip set_cs_const).  The choice must follow the
    fields: constant -> one walker made τ-dependent -> constant again, through both upload entry points, 12 systems per launch (the
    LDS-resident kernel) and with a graph-captured solve in between (a captured graph holds the kernel variant)."""
    Lt, Lc = 10, 3
    m = lat.holstein_honeycomb(Lc, Lt)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    N, Nh, nw = m.fpi.N, nt.shape[1], 12
    h = L.Handle(Lt, N, nt, colors, True, nw, 1)
    g = np.random.default_rng(5)
    tb = 1.0 + 0.2 * g.standard_normal(Nh)  # per-bond hopping, the same on every slice

    def put(w, varying):
        V = np.asfortranarray(0.7 * g.standard_normal((N, Lt)))
        t = np.asfortranarray(np.repeat(tb[:, None], Lt, axis=1))
        if varying:
            t[3, Lt - 1] += 0.25  # one bond on one slice
        expV, ch, sh = orc.update_fields(V, t, perm, 0.05, True)
        if entry == "path_integral":
            h.call("smoqy_update_from_path_integral", w, L.ptr(V), L.ptr(t), L.ptr(perm), C.c_double(0.05))
        else:
            h.call("smoqy_update_fields", w, L.ptr(expV), L.ptr(ch), L.ptr(sh))
        return orc.OracleFDM(nt, expV, ch, sh, True)

    oracles = [put(w, False) for w in range(nw)]
    v = rand(Lt, N, nw, 3)
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)

    def check(tag):
        for op, name in ((L.OP_M, "mul_M"), (L.OP_MT, "mul_Mt"), (L.OP_MTM, "mul_MtM"), (L.OP_MMT, "mul_MMt")):
            h.call("smoqy_matvec_v", op, b, a)
            got = h.vec_download(b)
            for s in (0, 7, 11):
                assert relerr(got[:, :, s], getattr(oracles[s], name)(v[:, :, s])) < 1e-13, (tag, name, s)

    def solve_graph(tag):
        h.call("smoqy_cg_use_graph", 1)
        x, it, eps = solve(h, v, 1e-10, 2000, 0)
        h.call("smoqy_cg_use_graph", 0)
        for s in (0, 7, 11):
            r = v[:, :, s] - oracles[s].mul_MtM(x[:, :, s])
            assert np.linalg.norm(r) / np.linalg.norm(v[:, :, s]) < 5e-10, (tag, s)

    check("constant")
    solve_graph("constant")
    oracles[7] = put(7, True)
    check("walker 7 tau-dependent")
    solve_graph("walker 7 tau-dependent")
    oracles[7] = put(7, False)
    check("constant again")
    solve_graph("constant again")


@pytest.mark.parametrize("Lt", [2, 4, 5, 6, 10, 12, 14, 16, 18, 20, 24, 36, 40, 48, 50, 64, 80, 96, 100, 128, 200, 320, 512])  # 320, 512: more than one table entry per lane
@pytest.mark.parametrize("is_sym", [True, False])
def test_in_place_tau_fft_form(Lt, is_sym):
    """smoqy_tfft_form(1): the single-image τ-FFT (decimation-in-frequency passes forward, decimation-in-time back, radix 4 / 2 / 3 / 5,
    digit-reversed order absorbed at the global-memory side) against the oracle's FourierTransformer (FourierTransformer.jl:39-64),
    the KPM preconditioner built on it (KPMPreconditioner.jl:355-414, 488-550) and the fused CG kernels (iteration counts within one
    of the two-image form's).  Lτ = 14 has a factor 7: the call is accepted and the two-image form stays."""
    N = 24
    h, o, nt, colors = make(lat.chain_neighbor_table(N), Lt, N, is_sym, seed=3, nrhs=3, vscale=0.5)
    v = rand(Lt, N, 3, 4)
    rv = np.random.default_rng(5).standard_normal(N)
    P = orc.OracleKPM(o[0])
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    ft = orc.OracleFT(Lt, N)
    res = {}
    for form in (0, 1):
        h.call("smoqy_tfft_form", form)
        w = v.copy(order="F")
        h.call("smoqy_fft_forward", L.ptr(w), 0, 3)
        for s in range(3):
            assert relerr(w[:, :, s], ft.forward(v[:, :, s])) < 1e-13, (form, s)
        h.call("smoqy_fft_inverse", L.ptr(w), 0, 3)
        assert relerr(w, v) < 1e-13
        out = np.zeros_like(v)
        h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, 3)
        assert relerr(out[:, :, 2], P.apply(v[:, :, 2])) < 1e-10
        res[form] = solve(h, v, 1e-10, 5000, 1)
    (x0, it0, eps0), (x1, it1, eps1) = res[0], res[1]
    assert np.abs(it0 - it1).max() <= 1  # the two forms differ in rounding (different pass order): a residual that lands on the tolerance may take one more step
    assert relerr(x1, x0) < 1e-9 and eps1.max() < 1e-10
    xo, ito, _ = o[0].cg_solve(v[:, :, 0], precond=P, tol=1e-10, maxiter=5000)
    # within one step on the short solves; the ≈ 100-iteration solve at Lτ = 512 lands two steps early (97 / 96 against 99) with the solution
    # agreeing to 1e-12: different summation orders, as on the long SSH solves (test_gpu_bench_shape.py)
    assert abs(int(it1[0]) - ito) <= max(1, ito // 40) and relerr(x1[:, :, 0], xo) < 1e-8


@pytest.mark.parametrize("Lt, N, nrhs", [(80, 37, 5), (100, 50, 5), (200, 21, 5), (200, 256, 2), (100, 24, 33)])
def test_register_blocked_tau_fft_for_R_M_R_lengths(Lt, N, nrhs):
    """tfft_rb_kernel (round 4): Lτ = R·M·R (80 = 4·5·4, 100 = 5·4·5, 200 = 5·8·5) with the first and the last radix-R stage in the
    registers of R·M·SB = 320 lanes and ONE LDS pass — FourierTransformer.jl:39-64 and the fused BLAS-1 lines of cg_solve!
    (ConjugateGradient.jl:219-245) as in the other forms.  smoqy_describe says which form the handle
    runs; ragged last tiles (N = 37, 50, 21 are no multiples of the tile width), several systems, both plain transforms against
    the oracle's dense-unitary-checked FourierTransformer, and the CG solves (iteration counts within one of the oracle's)."""
    is_sym = True
    h, o, nt, colors = make(lat.chain_neighbor_table(N), Lt, N, is_sym, seed=11, nrhs=nrhs, vscale=0.5)
    # the rule (api_cg.hip, tfft_rb_rule): up to 64 systems per launch on a handle left at the library's default; below 32 once the caller
    # has asked for the in-place form (several handles share the GPU)
    assert h.describe()["tfft"].startswith("tfft_rb_kernel")
    h.call("smoqy_tfft_form", 1)
    assert h.describe()["tfft"].startswith("tfft_rb_kernel" if nrhs < 32 else "tfft_kernel (in-place"), h.describe()
    h.call("smoqy_tfft_form", 0)
    assert h.describe()["tfft"].startswith("tfft_rb_kernel")
    v = rand(Lt, N, nrhs, 9)
    ft = orc.OracleFT(Lt, N)
    w = v.copy(order="F")
    h.call("smoqy_fft_forward", L.ptr(w), 0, nrhs)
    for s in range(nrhs):
        assert relerr(w[:, :, s], ft.forward(v[:, :, s])) < 1e-13, s
    h.call("smoqy_fft_inverse", L.ptr(w), 0, nrhs)
    assert relerr(w, v) < 1e-13
    # a sub-range of systems (sys_first > 0)
    if nrhs >= 3:
        w = v.copy(order="F")
        h.call("smoqy_fft_forward", L.ptr(w[:, :, 1:]), 1, 2)
        assert relerr(w[:, :, 1], ft.forward(v[:, :, 1])) < 1e-13 and relerr(w[:, :, 2], ft.forward(v[:, :, 2])) < 1e-13
        assert np.array_equal(w[:, :, 0], v[:, :, 0])
    rv = np.random.default_rng(5).standard_normal(N)
    P = orc.OracleKPM(o[0])
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, nrhs)
    assert relerr(out[:, :, nrhs - 1], P.apply(v[:, :, nrhs - 1])) < 1e-10
    x, it, eps = solve(h, v, 1e-10, 5000, 1)
    for s in (0, nrhs - 1):
        xo, ito, _ = o[0].cg_solve(v[:, :, s], precond=P, tol=1e-10, maxiter=5000)
        assert abs(int(it[s]) - ito) <= max(1, ito // 40) and relerr(x[:, :, s], xo) < 1e-8, (s, int(it[s]), ito)
        r = v[:, :, s] - o[0].mul_MtM(x[:, :, s])
        assert np.linalg.norm(r) / np.linalg.norm(v[:, :, s]) < 1.05e-10


def test_register_blocked_tau_fft_is_not_taken_above_64_systems():
    h, o, nt, colors = make(lat.chain_neighbor_table(24), 100, 24, True, seed=11, nrhs=65, vscale=0.5)
    assert h.describe()["tfft"].startswith("tfft_kernel (in-place"), h.describe()  # 32 systems or more: the library's default is the in-place form


def test_speculative_solve_restarts_when_the_preconditioner_grows():
    """Round 3: a solve that can be restarted (x === b) is launched on the host's CURRENT knowledge of the preconditioner (how many
    frequencies carry a chain, whether any walker is active) while the status record of the update in front of it is still on its way,
    and starts over if the record says that knowledge was stale in a way that matters.  Provoked here: fields with a narrow spectrum
    (few multi-term frequencies), then fields with a wide one on the SAME handle — the second solve's first launches are sized for the
    narrow spectrum, its light workgroups meet chains and poison them, the first poll consumes the record and the solve restarts.  The
    result must be bit for bit what a fresh handle (which waits for its first record) gives on the wide-spectrum fields, and the
    oracle's iteration count."""
    m = lat.holstein_honeycomb(4, 40)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N, Nh = 40, 32, m.fpi.t.shape[0]
    g = np.random.default_rng(77)
    V = np.asfortranarray(0.3 * g.standard_normal((N, Lt)))
    t_narrow = np.asfortranarray(np.full((Nh, Lt), 0.15))
    t_wide = np.asfortranarray(np.full((Nh, Lt), 2.2))
    rv = g.standard_normal(N)
    b = rand(Lt, N, 1, 5)

    def heavy_count(h):
        act, norder = C.c_int(0), C.c_int(0)
        order = np.zeros(Lt, dtype=np.int32)
        h.call("smoqy_precond_get", 0, C.byref(act), None, order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(norder), None, None)
        return int(act.value), int((order[: norder.value] > 1).sum())

    def solve_xb(h):
        x = h.vec_alloc()
        h.vec_upload(x, b)
        it, eps = np.zeros(1, dtype=np.int32), np.zeros(1)
        h.call("smoqy_cg_solve_v", x, x, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
        out = h.vec_download(x)
        h.call("smoqy_vec_free", x)
        return out, int(it[0]), float(eps[0])

    # handle A: narrow spectrum first (its record is consumed by the first solve), then the wide one with the record still in flight
    hA = L.Handle(Lt, N, nt, colors, True, 1, 1)
    hA.call("smoqy_update_from_path_integral", 0, L.ptr(V), L.ptr(t_narrow), L.ptr(perm), C.c_double(0.05))
    hA.call("smoqy_precond_update", 0, L.ptr(rv))
    solve_xb(hA)
    act_n, heavy_n = heavy_count(hA)
    hA.call("smoqy_update_from_path_integral", 0, L.ptr(V), L.ptr(t_wide), L.ptr(perm), C.c_double(0.05))
    hA.call("smoqy_precond_update", 0, L.ptr(rv))   # no host synchronisation: the status record is pending
    xA, itA, epsA = solve_xb(hA)                    # speculates with the narrow-spectrum count, restarts
    act_w, heavy_w = heavy_count(hA)
    assert act_n == 1 and act_w == 1 and heavy_w > heavy_n, (heavy_n, heavy_w)
    # handle B: the wide spectrum from scratch (first solve of a handle waits for its record)
    hB = L.Handle(Lt, N, nt, colors, True, 1, 1)
    hB.call("smoqy_update_from_path_integral", 0, L.ptr(V), L.ptr(t_wide), L.ptr(perm), C.c_double(0.05))
    hB.call("smoqy_precond_update", 0, L.ptr(rv))
    xB, itB, epsB = solve_xb(hB)
    assert np.all(np.isfinite(xA)) and epsA < 1e-10
    assert itA == itB and np.array_equal(xA, xB)
    expV, ch, sh = orc.update_fields(V, t_wide, perm, 0.05, True)
    o = orc.OracleFDM(nt, expV, ch, sh, True)
    P = orc.OracleKPM(o)
    P.update(rv)
    xo, ito, _ = o.cg_solve(b[:, :, 0], precond=P, tol=1e-10, maxiter=10000)
    assert abs(itA - ito) <= 1 and relerr(np.asarray(xA).reshape(Lt, N, order="F"), xo) < 1e-8
    hA.close()
    hB.close()


def test_cg_iteration_timing_brackets_the_four_launches():
    """smoqy_cg_iteration_timing (bench.py's iteration_kernels): the next n full-batch fused iterations get an event in front of each of their
    four launches; the read returns four positive means, the number of iterations sampled, and ends the sampling"""
    from smoqyelphqmc_amd.walkers import WalkerBatch

    b = WalkerBatch("holstein_honeycomb_L4_Ltau40", nwalkers=2, cg_split=1)
    b.sweep()
    b.h.call("smoqy_cg_iteration_timing", 8)
    b.sweep()
    us = (C.c_double * 4)()
    n = C.c_int(0)
    b.h.call("smoqy_cg_iteration_timing_read", us, C.byref(n))
    assert n.value == 8 and all(0.5 < v < 500.0 for v in us)
    b.sweep()
    b.h.call("smoqy_cg_iteration_timing_read", us, C.byref(n))
    assert n.value == 0
    b.h.close()
