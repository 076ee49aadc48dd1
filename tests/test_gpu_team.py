"""Walker teams (include/smoqy_hip.h "walker teams", csrc/team.hip): K host threads, each running the per-walker update sequence of
the reference (one walker per MPI rank, tutorials/holstein_honeycomb_mpi.jl:60-72) against its own walker index, must get exactly what
the batched entry points give when one caller drives all K walkers in lock step — and what the oracle's restatement of the reference's
per-walker sequence gives for that walker (oracle/sweep.py; parity unpinned like everything that rests on the oracle)."""
import ctypes as C
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch, WalkerTeam
from oracle import oracle as orc
from oracle.sweep import OracleWalker

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,K", [("holstein_honeycomb_L4_Ltau40", 4), ("bssh_chain_L256_Ltau200_alpha0p2", 3)])
def test_team_members_equal_the_batched_calls(name, K):
    team = WalkerTeam(name, K)
    ref = WalkerBatch(name, nwalkers=K, device_efa=False)
    g = np.random.default_rng(11)
    Lt, N, Nph = ref.Lt, ref.N, ref.Nph_force
    Rs = np.asfortranarray((g.standard_normal((Lt, N, K)) + 1j * g.standard_normal((Lt, N, K))) * np.sqrt(0.5))
    xs = np.array(ref.xs_force, copy=True)
    xs[:, :, : ref.Nph] += 0.05 * g.standard_normal((K, Lt, ref.Nph))
    rvs = np.ascontiguousarray(g.standard_normal((K, N)))
    # reference: one caller, batched entry points
    ref.h.vec_upload(ref.phi, Rs)
    rr_ref = ref.h.vec_dot(ref.phi, ref.phi).real
    ref.h.call("smoqy_matvec_v", L.OP_MT, ref.phi, ref.phi)
    ref.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, ref.phi, ref.phi)
    sf, it, ep = np.zeros(K), np.zeros(K, dtype=np.int32), np.zeros(K)
    dS = np.zeros((K, Lt, Nph))
    ref.h.call("smoqy_pff_step_v", ref.phi, ref.u, L.ptr(xs), L.ptr(rvs), C.c_double(1e-10), 10000, 1, L.ptr(sf), L.ptr(it), L.ptr(ep), L.ptr(dS))
    # team: K threads, each with its own walker only, arriving in scrambled order
    out = [None] * K

    def member(w):
        m = team.members[w]
        threading.Event().wait(0.01 * ((w * 7) % K))
        rr = C.c_double(0.0)
        R = np.asfortranarray(Rs[:, :, w])
        team.call("smoqy_team_sample_phi", w, L.ptr(R), C.byref(rr))
        s, e, i = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        d = np.zeros((Lt, Nph))
        x = np.ascontiguousarray(xs[w])
        team.call("smoqy_team_pff_step", w, L.ptr(x), L.ptr(np.ascontiguousarray(rvs[w])), C.c_double(1e-10), 10000, 1, C.byref(s), C.byref(i), C.byref(e), L.ptr(d))
        # a second round with x unchanged (NULL) and no force
        s2, e2, i2 = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        team.call("smoqy_team_pff_step", w, None, L.ptr(np.ascontiguousarray(rvs[w])), C.c_double(1e-10), 10000, 1, C.byref(s2), C.byref(i2), C.byref(e2), None)
        out[w] = (rr.value, s.value, i.value, e.value, d, s2.value, i2.value)
        assert m.w == w

    with ThreadPoolExecutor(K) as pool:
        list(pool.map(member, range(K)))
    for w in range(K):
        rr, s, i, e, d, s2, i2 = out[w]
        assert rr == rr_ref[w] and s == sf[w] and i == it[w] and e == ep[w]  # same kernels on the same batch: bit for bit
        assert np.array_equal(d, dS[w])
        assert i2 == i and abs(s2 - s) <= 1e-9 * abs(s)  # same fields, a fresh Lanczos with the same start vector: the same solve
    # and against the oracle, walker by walker: Φ = Λᵀ M† R on the initial fields (src/PFFCalculator.jl:67-73), then S_f and ∂S_f/∂x on
    # the fields of xs (:79-116, :146-155)
    for w in range(K):
        ow = OracleWalker(name, walker=w)
        assert np.array_equal(ow.x, ref.xs_force[w].T)
        ow.phi = orc.lambda_apply(ow.Lam, ow.fdm.mul_Mt(np.asfortranarray(Rs[:, :, w])), "mulT")
        ow.x[...] = xs[w].T
        ow.refresh_fields()
        s_o, it_o, eps_o = ow.action(1e-10, rvs[w])
        dS_o = ow.force()
        rr, s, i, e, d, s2, i2 = out[w]
        assert abs(rr - np.vdot(Rs[:, :, w], Rs[:, :, w]).real) < 1e-12 * rr
        assert abs(i - it_o) <= 1 and e < 1e-10 and eps_o < 1e-10
        assert abs(s - s_o) < 1e-8 * abs(s_o), (s, s_o)
        assert np.abs(d.T - dS_o).max() < 1e-7 * np.abs(dS_o).max(), np.abs(d.T - dS_o).max()
    team.close()
    ref.h.close()


def test_team_sweeps_and_errors():
    K = 4
    team = WalkerTeam("holstein_honeycomb_L4_Ltau40", K)
    with ThreadPoolExecutor(K) as pool:
        res = list(pool.map(lambda m: m.sweep(), team.members))
    assert all(r[1] > 0 and r[2] < 1e-10 for r in res)
    # a member that never arrives: the others time out with an error instead of hanging
    team.call("smoqy_team_set_timeout", C.c_double(0.5))
    errs = []

    def lonely(w):
        try:
            team.members[w].sample_pseudofermion_fields()
        except L.SmoqyError as ex:
            errs.append(str(ex))

    with ThreadPoolExecutor(K - 1) as pool:
        list(pool.map(lonely, range(K - 1)))
    assert len(errs) == K - 1 and all("timed out" in e for e in errs)
    # the team is usable again afterwards
    team.call("smoqy_team_set_timeout", C.c_double(600.0))
    with ThreadPoolExecutor(K) as pool:
        rr = list(pool.map(lambda m: m.sample_pseudofermion_fields(), team.members))
    assert all(r > 0 for r in rr)
    team.close()


def test_native_member_threads_run_the_member_sweep():
    """smoqy_team_bench_sweeps (measurement aid of bench.py's team scan): K std::threads inside the library run the member sweep through the
    team entry points.  Its counts are fixed by the sweep's shape — 3 + Nt solves per member and sweep — and the solves converge in the
    iteration range the Python members see on the same lattice."""
    K = 3
    team = WalkerTeam("holstein_honeycomb_L4_Ltau40", K)
    b = team.batch
    x0 = np.ascontiguousarray(np.stack([np.asarray(b.xs_force[w]) for w in range(K)]))
    secs, so, itn = C.c_double(0.0), C.c_long(0), C.c_long(0)
    team.call("smoqy_team_bench_sweeps", L.ptr(x0), int(b.Nph), C.c_double(b.drift), int(b.Nt), C.c_double(b.tol), C.c_double(b.tol_force), int(b.maxiter), 0, 1, 2, 99,
              C.byref(secs), C.byref(so), C.byref(itn))
    assert so.value == K * 2 * (3 + b.Nt) and secs.value > 0
    native = itn.value / so.value
    with ThreadPoolExecutor(K) as pool:
        list(pool.map(lambda m: m.sweep(), team.members))
    py = sum(m.iters_sum for m in team.members) / sum(m.solves for m in team.members)
    assert abs(native - py) <= 0.25 * py + 2, (native, py)
    with pytest.raises(L.SmoqyError):
        team.call("smoqy_team_bench_sweeps", L.ptr(x0), int(b.Nph_force) + 1, C.c_double(b.drift), int(b.Nt), C.c_double(b.tol), C.c_double(b.tol_force), int(b.maxiter), 0, 0, 1, 99,
                  C.byref(secs), C.byref(so), C.byref(itn))
    team.close()


def test_members_in_other_processes_equal_the_batched_calls(tmp_path):
    """smoqy_team_serve / smoqy_member_*: the members are separate PROCESSES (the reference's MPI ranks) that never touch the GPU; each
    joins the published team through shared memory, sends its own arrays and must receive exactly what one caller gets from the batched
    entry points for the same inputs."""
    import json
    import os
    import subprocess
    import sys

    name, K, seed = "holstein_honeycomb_L4_Ltau40", 3, 20261004
    team = WalkerTeam(name, K)
    ref = WalkerBatch(name, nwalkers=K, device_efa=False)
    info = dict(team.serve(f"/smoqy-test-{os.getpid()}"), seed=seed)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_team_member_child.py")
    procs = [subprocess.Popen([sys.executable, child, json.dumps(info), str(w), "parity", str(tmp_path / f"m{w}.npz")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(K)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    # an in-process call is refused while the team is published
    with pytest.raises(L.SmoqyError):
        team.call("smoqy_team_sample_phi", 0, L.ptr(np.zeros((ref.Lt, ref.N), dtype=np.complex128, order="F")), C.byref(C.c_double(0.0)))
    team.unserve()
    Lt, N, Nph = ref.Lt, ref.N, ref.Nph_force
    Rs = np.empty((Lt, N, K), dtype=np.complex128, order="F")
    xs = np.array(ref.xs_force, copy=True)
    rvs = np.empty((K, N))
    for w in range(K):
        g = np.random.default_rng([seed, w])
        Rs[:, :, w] = (g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N))) * np.sqrt(0.5)
        xs[w, :, : ref.Nph] += 0.05 * g.standard_normal((Lt, ref.Nph))
        rvs[w] = g.standard_normal(N)
    ref.h.vec_upload(ref.phi, Rs)
    rr_ref = ref.h.vec_dot(ref.phi, ref.phi).real
    ref.h.call("smoqy_matvec_v", L.OP_MT, ref.phi, ref.phi)
    ref.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, ref.phi, ref.phi)
    sf, it, ep = np.zeros(K), np.zeros(K, dtype=np.int32), np.zeros(K)
    dS = np.zeros((K, Lt, Nph))
    ref.h.call("smoqy_pff_step_v", ref.phi, ref.u, L.ptr(xs), L.ptr(rvs), C.c_double(1e-10), 10000, 1, L.ptr(sf), L.ptr(it), L.ptr(ep), L.ptr(dS))
    for w in range(K):
        r = np.load(tmp_path / f"m{w}.npz")
        assert float(r["rr"]) == rr_ref[w] and float(r["sf"]) == sf[w] and int(r["it"]) == it[w] and float(r["eps"]) == ep[w]
        assert np.array_equal(r["dS"], dS[w])
    team.close()



def _reference_trajectory(name, K, second=False):
    """one caller, batched entry points, the same per-walker random streams as the members: (ΔH, proposed fields, original fields[, ΔH of
    a second trajectory after the even walkers accepted and the odd ones rejected])"""
    ref = WalkerBatch(name, nwalkers=K, device_efa=True)
    x_before = np.array(ref.xs_force, copy=True)
    dH, _ = ref.hmc_trajectory_device()
    x_prop = np.zeros_like(x_before)
    ref.h.call("smoqy_efa_get_state", L.ptr(x_prop), None)
    dH2 = None
    if second:
        ref.h.call("smoqy_efa_restore_walkers", L.ptr(np.array([w % 2 for w in range(K)], dtype=np.int32)))
        dH2 = np.array(ref.hmc_trajectory_device()[0], copy=True)
    ref.h.close()
    return (np.array(dH, copy=True), x_prop, x_before) + ((dH2,) if second else ())


@pytest.mark.parametrize("name,K", [("holstein_honeycomb_L4_Ltau40", 4), ("bssh_chain_L256_Ltau200_alpha0p2", 3)])
def test_team_hmc_update_equals_the_batched_trajectory(name, K):
    """smoqy_team_hmc_update / smoqy_team_hmc_finish: every member's hmc_update! (src/EFAPFFHMCUpdater.jl:102-276) runs its trajectory on the
    device, all members in one batched call; ΔH and the proposed fields must be those of one caller driving the batched entry points with
    the same random streams, and each member's own accept / reject decision must leave its walker — and only its walker — accordingly."""
    dH_ref, x_prop, x_before, dH2_ref = _reference_trajectory(name, K, second=True)
    team = WalkerTeam(name, K, device_efa=True)
    out = [None] * K
    out2 = [None] * K

    def member(w):
        m = team.members[w]
        threading.Event().wait(0.01 * ((w * 5) % K))
        dH, x_new = m.hmc_update()
        m.hmc_finish(w % 2 == 0, x_new)
        out[w] = (dH, x_new)

    with ThreadPoolExecutor(K) as pool:
        list(pool.map(member, range(K)))
    x_dev = np.zeros_like(x_before)
    team.batch.h.call("smoqy_efa_get_state", L.ptr(x_dev), None)
    for w in range(K):
        assert out[w][0] == dH_ref[w]                       # same kernels on the same batch: bit for bit
        assert np.array_equal(out[w][1], x_prop[w])
        assert np.array_equal(x_dev[w], x_prop[w] if w % 2 == 0 else x_before[w])
        assert np.array_equal(team.members[w].x, x_dev[w])
    # the calls must come in pairs
    with pytest.raises(L.SmoqyError):
        with ThreadPoolExecutor(K) as pool:
            list(pool.map(lambda w: team.members[w].hmc_finish(True), range(K)))

    # a second update WITHOUT sending x: every walker continues from where its own decision left it on the device
    def member2(w):
        m = team.members[w]
        dH, x_new = m.hmc_update(send_x=False)
        m.hmc_finish(False)
        out2[w] = dH

    with ThreadPoolExecutor(K) as pool:
        list(pool.map(member2, range(K)))
    assert all(out2[w] == dH2_ref[w] for w in range(K))
    # a following per-step round without x sees the fields the decisions left
    with ThreadPoolExecutor(K) as pool:
        list(pool.map(lambda w: team.members[w].sample_pseudofermion_fields(), range(K)))
        res = list(pool.map(lambda w: team.members[w].pff_step(1e-10, moved=False, want_force=False), range(K)))
    assert all(np.isfinite(r[0]) and r[2] < 1e-10 for r in res)
    team.close()


def test_hmc_update_of_members_in_other_processes(tmp_path):
    """the same through smoqy_team_serve / smoqy_member_hmc_update: the members are processes without GPU access"""
    import json
    import os
    import subprocess
    import sys

    name, K = "holstein_honeycomb_L4_Ltau40", 3
    dH_ref, x_prop, x_before = _reference_trajectory(name, K)
    team = WalkerTeam(name, K, device_efa=True)
    info = dict(team.serve(f"/smoqy-test-hmc-{os.getpid()}"), seed=0)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_team_member_child.py")
    procs = [subprocess.Popen([sys.executable, child, json.dumps(info), str(w), "hmc", str(tmp_path / f"h{w}.npz")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(K)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    team.unserve()
    x_dev = np.zeros_like(x_before)
    team.batch.h.call("smoqy_efa_get_state", L.ptr(x_dev), None)
    for w in range(K):
        r = np.load(tmp_path / f"h{w}.npz")
        assert float(r["dH"]) == dH_ref[w]
        assert np.array_equal(r["x_new"], x_prop[w])
        assert np.array_equal(x_dev[w], x_prop[w] if w % 2 == 0 else x_before[w])
    team.close()


def test_team_error_paths_leave_the_team_usable():
    """a member that never arrives (code 9 after the team's time-out), members that make different calls in one round (code 8), a published
    team that is withdrawn while a member waits (code 10): every waiting member gets an error instead of hanging, and the team works
    afterwards"""
    import os

    from smoqyelphqmc_amd.walkers import RemoteMember

    K = 2
    team = WalkerTeam("holstein_honeycomb_L4_Ltau40", K)
    b = team.batch
    R = np.asfortranarray(np.random.default_rng(1).standard_normal((b.Lt, b.N)) + 0j)
    rr = C.c_double(0.0)
    team.call("smoqy_team_set_timeout", C.c_double(0.3))
    lib = team.lib
    # (9) the other member never comes
    assert lib.smoqy_team_sample_phi(team._t, 0, L.ptr(R), C.byref(rr)) == 9
    assert b"timed out" in lib.smoqy_team_last_error(team._t)
    # (8) different calls in one round: the late caller is refused, the early one completes when its partner makes the right call
    team.call("smoqy_team_set_timeout", C.c_double(20.0))
    res = {}

    def early():
        res["early"] = lib.smoqy_team_sample_phi(team._t, 0, L.ptr(R), C.byref(rr))

    th = threading.Thread(target=early)
    th.start()
    threading.Event().wait(0.2)
    sf, ep, it = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
    rv = np.random.default_rng(2).standard_normal(b.N)
    x = np.ascontiguousarray(b.xs_force[1])
    assert lib.smoqy_team_pff_step(team._t, 1, L.ptr(x), L.ptr(rv), C.c_double(1e-8), 1000, 1, C.byref(sf), C.byref(it), C.byref(ep), None) == 8
    rr1 = C.c_double(0.0)
    assert lib.smoqy_team_sample_phi(team._t, 1, L.ptr(R), C.byref(rr1)) == 0
    th.join(30)
    assert res["early"] == 0 and rr.value == rr1.value > 0
    # (10) a published team withdrawn under a waiting member
    info = team.serve(f"/smoqy-test-err-{os.getpid()}")
    m0 = RemoteMember(info, 0)
    with pytest.raises(L.SmoqyError):  # index 0 is taken
        RemoteMember(info, 0, wait_seconds=1.0)

    def waiting():
        try:
            m0.sample_pseudofermion_fields()
            res["remote"] = 0
        except L.SmoqyError as e:
            res["remote"] = str(e)

    th = threading.Thread(target=waiting)
    th.start()
    threading.Event().wait(0.3)
    team.unserve()
    th.join(30)
    assert "(10)" in res["remote"] and "withdrawn" in res["remote"]
    m0.close()
    # the in-process entry points work again
    with ThreadPoolExecutor(K) as pool:
        rrs = list(pool.map(lambda w: team.members[w].sample_pseudofermion_fields(), range(K)))
    assert all(v > 0 for v in rrs)
    team.close()


def test_a_rank_that_dies_does_not_wedge_the_others(tmp_path):
    """a member PROCESS is killed while it waits in a round (an MPI rank that crashes): the surviving member gets the rendezvous time-out
    (code 9) instead of hanging, the dead rank's index can be attached again, and the next complete round works"""
    import json
    import os
    import signal
    import subprocess
    import sys
    import time

    from smoqyelphqmc_amd.walkers import RemoteMember

    K = 2
    team = WalkerTeam("holstein_honeycomb_L4_Ltau40", K)
    team.call("smoqy_team_set_timeout", C.c_double(1.5))
    info = dict(team.serve(f"/smoqy-test-kill-{os.getpid()}"), seed=3)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_team_member_child.py")
    # rank 1 joins and blocks in its first call (rank 0 has not called yet) ... and is killed there
    p = subprocess.Popen([sys.executable, child, json.dumps(info), "1", "sweeps", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(2.0)
    p.send_signal(signal.SIGKILL)
    p.wait(timeout=30)
    m0 = RemoteMember(info, 0)
    # if the dead rank had already arrived in its round, that round completes with its last deposit (otherwise it times out like the next) ...
    try:
        assert m0.sample_pseudofermion_fields() > 0
    except L.SmoqyError as first:
        assert "(9)" in str(first)
    # ... and the next one cannot: the survivor gets the time-out, it does not hang
    t0 = time.perf_counter()
    with pytest.raises(L.SmoqyError) as e:
        m0.pff_step(1e-8, moved=True, want_force=False)
    assert "(9)" in str(e.value) and time.perf_counter() - t0 < 10.0
    # a restarted rank takes the dead one's index over (its pid is gone), and complete rounds work again
    m1 = RemoteMember(info, 1)
    res = {}

    def other():
        res["rr1"] = m1.sample_pseudofermion_fields()
        res["s1"] = m1.pff_step(1e-8, moved=True, want_force=False)

    th = threading.Thread(target=other)
    th.start()
    rr0 = m0.sample_pseudofermion_fields()
    s0 = m0.pff_step(1e-8, moved=True, want_force=False)
    th.join(60)
    assert rr0 > 0 and res["rr1"] > 0 and s0[2] < 1e-8 and res["s1"][2] < 1e-8
    m0.close()
    m1.close()
    team.unserve()
    team.close()


def test_team_greens_estimator_equals_the_batched_measurement(tmp_path):
    """smoqy_team_ge_update / smoqy_team_ge_measure_GD0 (and smoqy_member_*): each member's update_greens_estimator! + measure_GΔ0!
    (src/Measurements/GreensEstimator.jl:125-233) — its own random vectors, all K·Nrv systems in one batched CG — must give the G(Δ,0) array
    and the iteration counts of one caller driving the batched entry points (WalkerBatch.measure_greens) with the same random streams:
    members as threads, then members as processes."""
    import json
    import os
    import subprocess
    import sys

    name, K, Nrv = "holstein_honeycomb_L4_Ltau40", 3, 4
    ref = WalkerBatch(name, nwalkers=K)
    G_ref, it_ref = ref.measure_greens(Nrv, orbitals=(1, 2))
    it_ref = it_ref.reshape(K, Nrv).sum(axis=1)
    ref._ge[0].close()
    ref.h.close()
    # threads
    team = WalkerTeam(name, K)
    team.ge_config(Nrv)
    with ThreadPoolExecutor(K) as pool:
        res = list(pool.map(lambda m: m.measure_greens(orbitals=(1, 2)), team.members))
    for w in range(K):
        assert np.array_equal(res[w][0], G_ref[w]) and res[w][1] == it_ref[w]
    team.close()
    # processes
    team = WalkerTeam(name, K)
    team.ge_config(Nrv)
    info = dict(team.serve(f"/smoqy-test-ge-{os.getpid()}"), seed=0)
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_team_member_child.py")
    procs = [subprocess.Popen([sys.executable, child, json.dumps(info), str(w), "ge", str(tmp_path / f"g{w}.npz")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(K)]
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    team.close()
    for w in range(K):
        r = np.load(tmp_path / f"g{w}.npz")
        assert np.array_equal(r["G"], G_ref[w]) and int(r["it"]) == it_ref[w]
    # G(r = 0, τ = 0) + G(r = 0, τ = β) = 1 for equal orbitals is checked by tests/test_gpu_greens.py on the batched path this one equals


def test_members_that_draw_their_random_numbers_one_sweep_ahead_run_the_same_sweeps():
    """TeamMember.prefetch_randoms(pool): the member's generator is asked for the arrays of the next sweep_device_hmc by a pool thread
    while the member waits in the rendezvous — the same arrays in the same order, hence the same ΔH and iteration counts as members that
    draw on demand.  The native driver draws the same way (one producer thread per member); its counts are fixed by the sweep's shape."""
    name, K, Nt = "holstein_honeycomb_L4_Ltau40", 3, 4
    ta = WalkerTeam(name, K, device_efa=True, Nt=Nt)
    tb = WalkerTeam(name, K, device_efa=True, Nt=Nt)
    draw_pool = ThreadPoolExecutor(2)
    for m in tb.members:
        m.prefetch_randoms(draw_pool)
    with ThreadPoolExecutor(2 * K) as pool:
        for _ in range(3):
            fa = [pool.submit(m.sweep_device_hmc) for m in ta.members]
            fb = [pool.submit(m.sweep_device_hmc) for m in tb.members]
            da, db = [f.result() for f in fa], [f.result() for f in fb]
            assert da == db
    assert [m.iters_sum for m in ta.members] == [m.iters_sum for m in tb.members]
    draw_pool.shutdown(wait=True)
    b = tb.batch
    x0 = np.ascontiguousarray(np.stack([np.asarray(b.xs_force[w]) for w in range(K)]))
    secs, so, itn = C.c_double(0.0), C.c_long(0), C.c_long(0)
    tb.call("smoqy_team_bench_sweeps", L.ptr(x0), int(b.Nph), C.c_double(b.drift), Nt, C.c_double(b.tol), C.c_double(b.tol_force), int(b.maxiter), 1, 1, 2, 5,
            C.byref(secs), C.byref(so), C.byref(itn))
    assert so.value == K * 2 * (3 + Nt) and itn.value > so.value and secs.value > 0
    ta.close()
    tb.close()


@pytest.mark.parametrize("published", [False, True])
def test_a_deadline_that_passes_inside_a_running_round_is_not_a_time_out(published):
    """the rendezvous deadline covers the wait for the OTHER MEMBERS only.  Here every member has arrived long before the
    deadline, but the round itself (a whole hmc_update! of 16 leapfrog steps on the headline lattice) runs far longer than the 0.05 s the
    team allows: the early members used to give up mid-round with code 9 while the round was still writing through their slots.  Now they
    get the round's result; in-process threads and members attached through shared memory alike."""
    import os

    from smoqyelphqmc_amd.walkers import RemoteMember

    K = 3
    team = WalkerTeam("holstein_honeycomb_L16_Ltau128", K, device_efa=True, Nt=16)
    team.call("smoqy_team_set_timeout", C.c_double(0.05))
    if published:
        info = team.serve(f"/smoqy-test-deadline-{os.getpid()}")
        members = [RemoteMember(info, w, seed=5) for w in range(K)]
        for w, m in enumerate(members):
            m.rng = np.random.Generator(np.random.PCG64(1000 + w))
    else:
        members = team.members
    with ThreadPoolExecutor(K) as pool:
        res = list(pool.map(lambda m: m.hmc_update(), members))          # raises SmoqyError on any non-zero code
        list(pool.map(lambda q: members[q].hmc_finish(False, res[q][1]), range(K)))
    for dH, x_new in res:
        assert np.isfinite(dH) and np.all(np.isfinite(x_new))
    # a member that really is alone still times out (the deadline is not gone)
    lib = team.lib
    if not published:
        R = np.asfortranarray(np.random.default_rng(1).standard_normal((team.batch.Lt, team.batch.N)) + 0j)
        rr = C.c_double(0.0)
        assert lib.smoqy_team_sample_phi(team._t, 0, L.ptr(R), C.byref(rr)) == 9
    else:
        with pytest.raises(L.SmoqyError) as e:
            members[0].sample_pseudofermion_fields()
        assert "(9)" in str(e.value)
        for m in members:
            m.close()
        team.unserve()
    team.close()


def test_complex_handles_are_refused_by_teams():
    """team staging holds N doubles per Lanczos start vector, a handle with T = ComplexF64 reads 2N — such handles are
    refused at smoqy_team_create with a message instead of being mis-staged."""
    import smoqyelphqmc_amd as sq

    m = sq.lattice.bssh_chain(8, 6)
    nt, perm, colors = sq.lattice.checkerboard_decomposition(m.fpi.neighbor_table)
    h = L.Handle(6, 8, nt, colors, True, 2, 1, -1, is_complex=True)
    lib = L.load()
    t = C.c_void_p()
    rc = lib.smoqy_team_create(C.byref(t), h._h, 8)
    assert rc != 0 and not t
    assert b"complex" in lib.smoqy_team_last_error(None)
    h.close()
