"""EFA leapfrog on the device against oracle/efa.py, and the one-call trajectory against the same trajectory
assembled step by step from entry points that have their own oracle tests.  PARITY UNPINNED: SmoQyDQMC's ExactFourierAccelerator is not
part of the reference tree; the oracle restates the published algorithm and the reference's call sites (src/EFAPFFHMCUpdater.jl:130-250)
and is pinned by tests/test_oracle_efa.py.  The last test is physics: leapfrog energy errors scale as Δt², which only holds when the
force the kernels produce is the gradient of the action the solves evaluate."""
import ctypes as C

import numpy as np
import pytest

from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle import efa

pytestmark = pytest.mark.gpu

WORKLOADS = ["holstein_honeycomb_L4_Ltau40", "bssh_chain_L256_Ltau200"]  # the second has a frozen (infinite-mass) partner mode


def host_state(b):
    x = np.zeros((b.nw, b.Lt, b.Nph_force))
    p = np.zeros_like(x)
    b.h.call("smoqy_efa_get_state", L.ptr(x), L.ptr(p))
    return x, p


@pytest.mark.parametrize("name", WORKLOADS)
def test_momentum_energies_and_evolution_match_the_oracle(name):
    b = WalkerBatch(name, nwalkers=2, device_efa=True)
    q, m = b.efa_q, b.efa_m
    g = np.random.default_rng(1)
    R = np.ascontiguousarray(g.standard_normal((b.nw, b.Lt, b.Nph_force)))
    K = np.zeros(b.nw)
    b.h.call("smoqy_efa_initialize_momentum", L.ptr(R), L.ptr(K))
    x, p = host_state(b)
    for w in range(b.nw):
        po, Ko = efa.initialize_momentum(R[w].T, m)
        np.testing.assert_allclose(p[w].T, po, atol=1e-12 * np.abs(po).max())
        assert abs(K[w] - Ko) < 1e-12 * Ko
        np.testing.assert_array_equal(x[w], b.xs_force[w])
    K2, Sb = b.efa_energies()
    np.testing.assert_allclose(K2, K, rtol=1e-12)
    for w in range(b.nw):
        assert abs(Sb[w] - efa.bosonic_action(x[w].T, q, m)) < 1e-12 * Sb[w]
    # exact evolution, without and with a kick from a force left on the device
    b.h.call("smoqy_efa_evolve", C.c_double(0.31), C.c_double(0.0), 1)
    x1, p1 = host_state(b)
    for w in range(b.nw):
        xo, po = efa.evolve_eom(x[w].T, p[w].T, 0.31, q, m)
        np.testing.assert_allclose(x1[w].T, xo, atol=1e-12 * np.abs(xo).max())
        np.testing.assert_allclose(p1[w].T, po, atol=1e-12 * np.abs(po).max())
    K3, Sb3 = b.efa_energies()
    np.testing.assert_allclose(K3 + Sb3, K + Sb, rtol=1e-12)       # the harmonic flow conserves K + S_b
    b.sample_pseudofermion_fields()
    sf, it, eps, force = b.pff_step(1e-10, moved=False, want_force=True)   # fields were refreshed by the evolve call
    force = force.copy()
    b.h.call("smoqy_efa_evolve", C.c_double(0.2), C.c_double(0.05), 1)
    x2, p2 = host_state(b)
    for w in range(b.nw):
        xo, po = efa.evolve_eom(x1[w].T, p1[w].T, 0.2, q, m, force=force[w].T, kick=0.05)
        np.testing.assert_allclose(x2[w].T, xo, atol=1e-12 * np.abs(xo).max())
        np.testing.assert_allclose(p2[w].T, po, atol=1e-11 * np.abs(po).max())
    if b.Nph_force != b.Nph:   # the frozen partner mode never moves and never gets momentum
        assert np.all(x2[:, :, b.Nph:] == 0.0) and np.all(p2[:, :, b.Nph:] == 0.0)
    # reject branch: x restored, fields follow
    b.h.call("smoqy_efa_checkpoint", 0)
    b.h.call("smoqy_efa_evolve", C.c_double(0.4), C.c_double(0.0), 1)
    b.h.call("smoqy_efa_checkpoint", 1)
    x3, _ = host_state(b)
    np.testing.assert_array_equal(x3, x2)


@pytest.mark.parametrize("name,nw,Nt,in_place", [("holstein_honeycomb_L4_Ltau40", 2, 4, False), ("holstein_honeycomb_L16_Ltau128", 16, 2, False),
                                                 ("holstein_honeycomb_L16_Ltau128", 16, 2, True), ("ossh_square_L12_Ltau100_alpha0p2", 4, 2, False)])
def test_one_call_trajectory_equals_the_step_by_step_trajectory(name, nw, Nt, in_place):
    """smoqy_hmc_trajectory_v against the same sequence driven from the host through smoqy_pff_step_v (x uploaded, force downloaded every
    step) and the ORACLE's evolve_eom: identical solves on identical fields, so positions and momenta agree to rounding.  The second case is
    the batch shape bench.py drives (16 walkers of the headline lattice) — once more with the one-call side on the in-place τ-FFT
    (smoqy_tfft_form, what bench.py selects for its timed batches) against the step-by-step side on the two-image form: same iteration
    counts, same trajectory —, the last an SSH model (hoppings follow the phonons; at its weak-coupling point: the α = 1 solves take ≈ 520
    iterations, over which the two drivers' fields — formed on the device from x, or from an x that went through the host — differ by
    rounding and the counts by ±1, so exact equality of the counts is asked of the short solves only)."""
    dt = 0.11
    a = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt, tfft_in_place=True if in_place else None)
    bb = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
    g = np.random.default_rng(3)
    Rphi = np.asfortranarray((g.standard_normal((a.Lt, a.N, nw)) + 1j * g.standard_normal((a.Lt, a.N, nw))) * np.sqrt(0.5))
    R = np.ascontiguousarray(g.standard_normal((nw, a.Lt, a.Nph_force)))
    rv = np.ascontiguousarray(g.standard_normal((Nt, nw, a.N)))
    outs = []
    for b in (a, bb):
        b.h.vec_upload(b.phi, Rphi)
        b.h.call("smoqy_matvec_v", L.OP_MT, b.phi, b.phi)
        b.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, b.phi, b.phi)
        K = np.zeros(nw)
        b.h.call("smoqy_efa_initialize_momentum", L.ptr(R), L.ptr(K))
    # (1) one call
    sf = np.zeros((Nt, nw)); it = np.zeros((Nt, nw), dtype=np.int32); ep = np.zeros((Nt, nw))
    a.h.call("smoqy_hmc_trajectory_v", a.phi, a.u, Nt, C.c_double(dt), C.c_double(1e-11), 10000, 1, L.ptr(rv), L.ptr(sf), L.ptr(it), L.ptr(ep))
    xa, pa = host_state(a)
    # (2) step by step on the host with the oracle's leapfrog
    x, p = host_state(bb)
    q, m = bb.efa_q, bb.efa_m
    X = [x[w].T.copy() for w in range(nw)]
    P = [p[w].T.copy() for w in range(nw)]

    def push():
        for w in range(nw):
            bb.xs_force[w] = X[w].T
        return L.ptr(bb.xs_force)

    for w in range(nw):
        X[w], P[w] = efa.evolve_eom(X[w], P[w], dt / 2, q, m)
    sf2 = np.zeros((Nt, nw)); it2 = np.zeros((Nt, nw), dtype=np.int32)
    for t in range(Nt):
        s1, i1, e1 = np.zeros(nw), np.zeros(nw, dtype=np.int32), np.zeros(nw)
        bb.h.call("smoqy_pff_step_v", bb.phi, bb.u, push(), L.ptr(np.ascontiguousarray(rv[t])), C.c_double(1e-11), 10000, 1, L.ptr(s1), L.ptr(i1), L.ptr(e1), L.ptr(bb.dSdx))
        sf2[t], it2[t] = s1, i1
        for w in range(nw):
            X[w], P[w] = efa.evolve_eom(X[w], P[w], dt / 2 if t == Nt - 1 else dt, q, m, force=bb.dSdx[w].T, kick=dt)
    assert np.array_equal(it, it2)
    np.testing.assert_allclose(sf, sf2, rtol=1e-9)
    for w in range(nw):
        np.testing.assert_allclose(xa[w].T, X[w], atol=1e-9 * np.abs(X[w]).max())
        np.testing.assert_allclose(pa[w].T, P[w], atol=1e-9 * np.abs(P[w]).max())


@pytest.mark.parametrize("name,is_sym", [("holstein_honeycomb_L4_Ltau40", False), ("bssh_chain_L256_Ltau200", True)])
def test_leapfrog_energy_error_scales_with_the_square_of_the_step(name, is_sym):
    """ΔH of a trajectory of fixed length shrinks ~4x when the step is halved: the kernels' force is the gradient of the action the
    solves evaluate (fermionic part) and the momentum kick / harmonic evolution are consistent with it.  Run where the reference's force
    IS the exact gradient — the Asym form and the SSH branches; its Holstein-only Sym branch peels the outer checkerboard factor with
    `transposed = true` (src/fermion_det_matrix_dervative.jl:71-74) and is exact only up to O(Δτ²) colour commutators, which puts a
    step-independent floor under ΔH there (measured: 3e-4 .. 2e-3 at either step; tests/test_oracle_force.py pins that quirk)."""
    dHs = []
    for Nt in (8, 16):
        b = WalkerBatch(name, nwalkers=2, device_efa=True, Nt=Nt, tol=1e-13, is_sym=is_sym)
        b.tol_force = 1e-12
        dH, last = b.hmc_trajectory_device(dt=0.4 / Nt)
        dHs.append(np.abs(dH))
        b.h.close()
    ratio = dHs[0] / dHs[1]
    assert np.all(dHs[1] < dHs[0]) and np.all(ratio > 2.5) and np.all(ratio < 6.5), (dHs, ratio)


def test_sweeps_with_the_random_numbers_drawn_one_sweep_ahead_are_the_same_sweeps():
    """WalkerBatch(prefetch_randoms=True) lets the host-thread pool draw sweep n + 1's random numbers while the device runs sweep n: every
    walker's generator is asked for the same arrays in the same order, so actions, ΔH, iteration counts and fields are those of the plain
    sweep, bit for bit — and a handle closed with a fill task still pending waits for it."""
    name, nw = "holstein_honeycomb_L4_Ltau40", 3
    a = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=4, host_threads=2)
    b = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=4, host_threads=2, prefetch_randoms=True)
    assert b.prefetch_randoms
    for _ in range(3):
        la, lb = a.sweep(), b.sweep()
        np.testing.assert_array_equal(la[0], lb[0])
        np.testing.assert_array_equal(la[1], lb[1])
        np.testing.assert_array_equal(a.dH, b.dH)
        np.testing.assert_array_equal(a.xs, b.xs)
    assert a.stats.iters_sum == b.stats.iters_sum and a.stats.solves == b.stats.solves
    assert b._draws is None and b._draws_next is not None  # the fourth sweep's numbers are being drawn
    b.h.close()                                            # waits for them before the page-locked arrays go
    assert all(f.done() for f in b._draws_next[1])
    a.h.close()


def _drive(b, Rphi, R, rv, Nt, dt, tol):
    """One trajectory on batch b from given deviates; returns (Sf, iters, eps, x, p)."""
    nw = b.nw
    b.h.vec_upload(b.phi, Rphi)
    b.h.call("smoqy_matvec_v", L.OP_MT, b.phi, b.phi)
    b.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, b.phi, b.phi)
    K = np.zeros(nw)
    b.h.call("smoqy_efa_initialize_momentum", L.ptr(R), L.ptr(K))
    sf = np.zeros((Nt, nw)); it = np.zeros((Nt, nw), dtype=np.int32); ep = np.zeros((Nt, nw))
    b.h.call("smoqy_hmc_trajectory_v", b.phi, b.u, Nt, C.c_double(dt), C.c_double(tol), 10000, 1, L.ptr(rv), L.ptr(sf), L.ptr(it), L.ptr(ep))
    x, p = host_state(b)
    return sf, it, ep, x, p


def _async_counts(b):
    runs, misses = C.c_long(0), C.c_long(0)
    b.h.call("smoqy_hmc_async", -1, C.byref(runs), C.byref(misses))
    return runs.value, misses.value


@pytest.mark.parametrize("name,nw", [("holstein_honeycomb_L4_Ltau40", 3), ("holstein_honeycomb_L16_Ltau128", 16), ("ossh_square_L12_Ltau100_alpha0p2", 2), ("bssh_chain_L256_Ltau200_alpha0p2", 2)])
def test_asynchronous_trajectory_equals_the_polling_trajectory(name, nw):
    """smoqy_hmc_async (round 4): from its second trajectory on a handle launches every force solve on the iteration count the step needed
    last time (+ margin) and waits for nothing until the end, where every solve is verified.  Same kernels, same iterations: positions,
    momenta, actions, iteration counts and residuals are IDENTICAL to the polling form's, trajectory after trajectory."""
    Nt, dt, tol = 6, 0.09, 1e-6
    a = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
    bb = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
    bb.h.call("smoqy_hmc_async", 0, None, None)
    g = np.random.default_rng(17)
    for b in (a, bb):
        b.h.call("smoqy_efa_checkpoint", 0)   # copyto!(x0, x): what every trajectory below is rejected back to
    for trip in range(4):
        Rphi = np.asfortranarray((g.standard_normal((a.Lt, a.N, nw)) + 1j * g.standard_normal((a.Lt, a.N, nw))) * np.sqrt(0.5))
        R = np.ascontiguousarray(g.standard_normal((nw, a.Lt, a.Nph_force)))
        rv = np.ascontiguousarray(g.standard_normal((Nt, nw, a.N)))
        outs = [_drive(b, Rphi, R, rv, Nt, dt, tol) for b in (a, bb)]
        for u, v in zip(*outs):
            assert np.array_equal(u, v), (trip, _async_counts(a), outs[0][1].T.tolist(), outs[1][1].T.tolist())
        assert np.all(outs[0][2] < tol) and np.all(outs[0][1] > 0)
        for b in (a, bb):   # the move is rejected: both start the next trajectory from the same fields
            b.h.call("smoqy_efa_checkpoint", 1)
    runs, misses = _async_counts(a)
    assert runs == 3 and misses == 0          # the first trajectory of a handle has no counts to launch on: it polls
    assert _async_counts(bb) == (0, 0)
    a.h.close(); bb.h.close()


def test_asynchronous_trajectory_falls_back_when_a_solve_needs_more_iterations():
    """The counts come from a trajectory on nearly free fields (x scaled down: the solves take a handful of iterations); the next one runs
    on the rough fields and needs three times as many — the check at the end of the asynchronous form finds unconverged solves, puts x, p
    and the fields back and repeats the trajectory with polls.  The caller sees the polling form's result, and a miss in the counters."""
    name, nw, Nt, dt, tol = "holstein_honeycomb_L4_Ltau40", 2, 5, 0.08, 1e-7
    a = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
    bb = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
    bb.h.call("smoqy_hmc_async", 0, None, None)
    g = np.random.default_rng(23)
    x_rough, p0 = host_state(a)
    iters = []
    for scale in (0.02, 1.0):
        Rphi = np.asfortranarray((g.standard_normal((a.Lt, a.N, nw)) + 1j * g.standard_normal((a.Lt, a.N, nw))) * np.sqrt(0.5))
        R = np.ascontiguousarray(g.standard_normal((nw, a.Lt, a.Nph_force)))
        rv = np.ascontiguousarray(g.standard_normal((Nt, nw, a.N)))
        outs = []
        for b in (a, bb):
            b.h.call("smoqy_efa_set_state", L.ptr(np.ascontiguousarray(scale * x_rough)), L.ptr(p0))
            outs.append(_drive(b, Rphi, R, rv, Nt, dt, tol))
        for u, v in zip(*outs):
            assert np.array_equal(u, v), scale
        iters.append(outs[0][1].max())
    assert iters[1] > iters[0] + 4, iters       # the premise: the second trajectory needs more than the margin covers
    runs, misses = _async_counts(a)
    assert (runs, misses) == (1, 1)
    a.h.close(); bb.h.close()
