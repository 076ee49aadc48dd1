"""The member side of a published walker team (smoqyelphqmc.jl_amd/csrc/member.cpp = libsmoqy_member.so: what a GPU-less rank of the
reference's one-walker-per-rank model links, tutorials/holstein_honeycomb_mpi.jl:60-72) against a CPU stand-in for the serving process
(tests/fake_team_server.cpp: same segment layout and protocol fields through team_shm.h, toy arithmetic).  No GPU: the GPU suite
(tests/test_gpu_team.py) covers the real server; here the protocol's corner cases run on every CPU check — index ownership and its
reclaim from a dead rank, the staging layout (a member's data goes through its own part of the segment only), the rendezvous deadline,
a round that outlasts the deadline, a server that dies inside a round, a withdrawn team, a failing round, mismatched calls."""
import ctypes as C
import os
import signal
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from smoqyelphqmc_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, LT, N, NPH = 3, 4, 6, 5
NX = NPH * LT


@pytest.fixture(scope="module")
def server_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fake_team") / "fake_team_server")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "fake_team_server.cpp"), "-pthread", "-lrt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.fixture(scope="module")
def lib():
    if os.environ.get("SMOQY_MEMBER_LIB"):   # the sanitizer child run below: the same entry points from an ASan + UBSan build of member.cpp
        L.MEMBER_LIB_PATH = os.environ["SMOQY_MEMBER_LIB"]
    return L.load_member()


class Server:
    def __init__(self, exe, mode="serve", timeout=0.4):
        self.name = f"/smoqy-fake-{os.getpid()}-{time.monotonic_ns()}"
        self.p = subprocess.Popen([exe, self.name, str(K), str(LT), str(N), str(NPH), str(timeout), mode], stdout=subprocess.PIPE, text=True)
        assert self.p.stdout.readline().strip() == "READY"

    def stop(self):
        if self.p.poll() is None:
            self.p.send_signal(signal.SIGTERM)
        self.p.wait(timeout=10)
        try:
            os.unlink("/dev/shm" + self.name)
        except OSError:
            pass


def attach(lib, name, w, wait=2.0):
    m = C.c_void_p()
    rc = lib.smoqy_member_attach(C.byref(m), name.encode(), w, C.c_double(wait))
    return rc, m


def in_threads(fns):
    """run one callable per member concurrently (ctypes releases the GIL inside the calls); returns their results in order"""
    out = [None] * len(fns)

    def run(i):
        out[i] = fns[i]()

    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
        assert not t.is_alive()
    return out


def member_data(w):
    g = np.random.default_rng(40 + w)
    return dict(R=g.standard_normal((LT * N, 2)), x=g.standard_normal(NX), rv=g.standard_normal(N), P=g.standard_normal(NX), rvs=g.standard_normal((LT + 1, N)))


def sample(lib, m, d):
    s = C.c_double(0)
    rc = lib.smoqy_member_sample_phi(m, L.ptr(d["R"]), C.byref(s))
    return rc, s.value


def pff(lib, m, d, tol=1e-8, force=True):
    Sf, it, eps, dS = C.c_double(0), C.c_int(0), C.c_double(0), np.zeros(NX)
    rc = lib.smoqy_member_pff_step(m, L.ptr(d["x"]), L.ptr(d["rv"]), C.c_double(tol), 100, 1, C.byref(Sf), C.byref(it), C.byref(eps), L.ptr(dS) if force else None)
    return rc, Sf.value, it.value, eps.value, dS


def test_attach_dims_fields_and_index_ownership(server_exe, lib):
    sv = Server(server_exe)
    try:
        rc, m0 = attach(lib, sv.name, 0)
        assert rc == 0
        rc, dup = attach(lib, sv.name, 0)
        assert rc == 1 and b"already attached" in lib.smoqy_member_last_error(None)
        rc, bad = attach(lib, sv.name, K)
        assert rc == 1 and b"outside the team" in lib.smoqy_member_last_error(None)
        dims = np.zeros(4, dtype=np.int32)
        assert lib.smoqy_member_dims(m0, dims.ctypes.data_as(C.POINTER(C.c_int))) == 0 and list(dims) == [LT, N, K, NPH]
        rc, m2 = attach(lib, sv.name, 2)
        assert rc == 0
        x = np.zeros(NX)
        assert lib.smoqy_member_fields(m2, L.ptr(x)) == 0
        np.testing.assert_array_equal(x, 200.0 + np.arange(NX))       # member 2's part of the field staging, nobody else's
        assert lib.smoqy_member_detach(m0) == 0
        rc, m0 = attach(lib, sv.name, 0)                                  # a detached index is free again
        assert rc == 0
        lib.smoqy_member_detach(m0)
        lib.smoqy_member_detach(m2)
    finally:
        sv.stop()


def test_index_of_a_dead_rank_is_reclaimed(server_exe, lib):
    sv = Server(server_exe)
    try:
        child = ("import sys, ctypes as C, os, time\nsys.path.insert(0, %r)\nfrom smoqyelphqmc_amd import _lib as L\nlib = L.load_member()\nm = C.c_void_p()\n"
                 "assert lib.smoqy_member_attach(C.byref(m), %r.encode(), 1, C.c_double(2.0)) == 0\nprint('ATTACHED', flush=True)\ntime.sleep(60)\n") % (ROOT, sv.name)
        p = subprocess.Popen([sys.executable, "-c", child], stdout=subprocess.PIPE, text=True)
        assert p.stdout.readline().strip() == "ATTACHED"
        rc, m = attach(lib, sv.name, 1)
        assert rc == 1                                                    # owned by a living rank
        p.kill()
        p.wait()
        rc, m = attach(lib, sv.name, 1)
        assert rc == 0                                                    # its owner is gone (ESRCH): the index is taken over
        lib.smoqy_member_detach(m)
    finally:
        sv.stop()


def test_rounds_carry_each_members_own_data(server_exe, lib):
    sv = Server(server_exe)
    try:
        ms = [attach(lib, sv.name, w)[1] for w in range(K)]
        ds = [member_data(w) for w in range(K)]
        for rc, s in in_threads([lambda w=w: sample(lib, ms[w], ds[w]) for w in range(K)]):
            assert rc == 0
        res = in_threads([lambda w=w: sample(lib, ms[w], ds[w]) for w in range(K)])
        for w, (rc, s) in enumerate(res):
            assert rc == 0 and np.isclose(s, (ds[w]["R"] ** 2).sum(), rtol=1e-14)
        res = in_threads([lambda w=w: pff(lib, ms[w], ds[w]) for w in range(K)])
        for w, (rc, Sf, it, eps, dS) in enumerate(res):
            assert rc == 0 and it == 10 + w and eps == 0.5e-8
            assert np.isclose(Sf, (ds[w]["x"] ** 2).sum() + ds[w]["rv"].sum(), rtol=1e-13)
            np.testing.assert_array_equal(dS, 2.0 * ds[w]["x"])
        # the trajectory call: Nt + 1 Lanczos start vectors per member, staged step-major across the team (team_shm.h, stage_in)
        Nt, dt = LT, 0.25

        def hmc(w):
            H0, H1, xn, it = np.zeros(3), np.zeros(3), np.zeros(NX), C.c_int(0)
            d = ds[w]
            rc = lib.smoqy_member_hmc_update(ms[w], L.ptr(d["x"]), L.ptr(d["R"]), L.ptr(d["P"]), L.ptr(np.ascontiguousarray(d["rvs"])), Nt, C.c_double(dt), C.c_double(1e-5),
                                             C.c_double(1e-8), 100, L.ptr(H0), L.ptr(H1), L.ptr(xn), C.byref(it))
            return rc, H0, H1, xn, it.value

        for w, (rc, H0, H1, xn, it) in enumerate(in_threads([lambda w=w: hmc(w) for w in range(K)])):
            d = ds[w]
            assert rc == 0 and it == Nt
            np.testing.assert_allclose(H0, [(d["P"] ** 2).sum(), (d["x"] ** 2).sum(), Nt], rtol=1e-13)
            np.testing.assert_allclose(H1, [d["rvs"].sum(), dt, 1e-5], rtol=1e-12)
            np.testing.assert_allclose(xn, d["x"] + dt * d["P"], rtol=1e-15)
        # Metropolis decisions per member: 0 and 2 accept, 1 rejects; smoqy_member_fields then shows who moved
        assert [r for r in in_threads([lambda w=w: lib.smoqy_member_hmc_finish(ms[w], int(w != 1)) for w in range(K)])] == [0, 0, 0]
        for w in range(K):
            x = np.zeros(NX)
            lib.smoqy_member_fields(ms[w], L.ptr(x))
            np.testing.assert_allclose(x, ds[w]["x"] + (dt * ds[w]["P"] if w != 1 else 0.0), rtol=1e-15)
        for m in ms:
            lib.smoqy_member_detach(m)
    finally:
        sv.stop()


def test_deadline_mismatch_and_failure_leave_the_team_usable(server_exe, lib):
    sv = Server(server_exe, timeout=0.6)
    try:
        ms = [attach(lib, sv.name, w)[1] for w in range(K)]
        ds = [member_data(w) for w in range(K)]
        # two of three members call: both time out (code 9) and take themselves out of the round
        t0 = time.monotonic()
        res = in_threads([lambda w=w: sample(lib, ms[w], ds[w]) for w in range(2)])
        assert [r[0] for r in res] == [9, 9] and 0.5 < time.monotonic() - t0 < 8
        assert b"timed out" in lib.smoqy_member_last_error(ms[0])
        # a member that makes a different call than the one already waiting is refused (8); the waiting one times out
        def late_pff():
            time.sleep(0.1)
            return pff(lib, ms[1], ds[1])[0]
        res = in_threads([lambda: sample(lib, ms[0], ds[0])[0], late_pff])
        assert res == [9, 8] and b"different calls" in lib.smoqy_member_last_error(ms[1])
        # a failing round reaches every member with the server's message
        res = in_threads([lambda w=w: pff(lib, ms[w], ds[w], tol=-1.0)[0] for w in range(K)])
        assert res == [7, 7, 7] and all(b"fake failure" in lib.smoqy_member_last_error(m) for m in ms)
        # and after all of that a complete round is served as if nothing had happened
        res = in_threads([lambda w=w: sample(lib, ms[w], ds[w]) for w in range(K)])
        for w, (rc, s) in enumerate(res):
            assert rc == 0 and np.isclose(s, (ds[w]["R"] ** 2).sum(), rtol=1e-14)
        for m in ms:
            lib.smoqy_member_detach(m)
    finally:
        sv.stop()


def test_a_running_round_is_not_subject_to_the_deadline(server_exe, lib):
    """the rendezvous deadline covers the wait for the other members only; the stand-in's rounds take three deadlines"""
    sv = Server(server_exe, mode="slow", timeout=0.25)
    try:
        ms = [attach(lib, sv.name, w)[1] for w in range(K)]
        ds = [member_data(w) for w in range(K)]
        t0 = time.monotonic()
        res = in_threads([lambda w=w: sample(lib, ms[w], ds[w]) for w in range(K)])
        assert time.monotonic() - t0 > 0.7
        for w, (rc, s) in enumerate(res):
            assert rc == 0 and np.isclose(s, (ds[w]["R"] ** 2).sum(), rtol=1e-14)
        for m in ms:
            lib.smoqy_member_detach(m)
    finally:
        sv.stop()


def test_server_that_dies_inside_a_round_and_a_withdrawn_team(server_exe, lib):
    sv = Server(server_exe, mode="die", timeout=0.2)
    try:
        ms = [attach(lib, sv.name, w)[1] for w in range(K)]
        ds = [member_data(w) for w in range(K)]
        reaper = threading.Thread(target=sv.p.wait)                      # a zombie still answers kill(pid, 0): the parent has to reap it
        reaper.start()
        res = in_threads([lambda w=w: sample(lib, ms[w], ds[w])[0] for w in range(K)])
        reaper.join(10)
        assert sv.p.returncode == 3
        assert res == [10, 10, 10] and b"died inside a round" in lib.smoqy_member_last_error(ms[0])
        for m in ms:
            lib.smoqy_member_detach(m)
    finally:
        sv.stop()
    sv = Server(server_exe, timeout=5.0)
    try:
        ms = [attach(lib, sv.name, w)[1] for w in range(K)]
        ds = [member_data(w) for w in range(K)]
        # one member waits for the others; the serving process withdraws the team: the waiter is released with code 10, later calls too
        killer = threading.Timer(0.3, lambda: sv.p.send_signal(signal.SIGTERM))
        killer.start()
        t0 = time.monotonic()
        rc, _ = sample(lib, ms[0], ds[0])
        assert rc == 10 and time.monotonic() - t0 < 4 and b"withdrawn" in lib.smoqy_member_last_error(ms[0])
        assert sample(lib, ms[1], ds[1])[0] == 10
        for m in ms:
            lib.smoqy_member_detach(m)
    finally:
        sv.stop()


def test_member_protocol_under_asan_and_ubsan(tmp_path):
    """member.cpp rebuilt with AddressSanitizer + UBSan and the tests above re-run against that build in a child process (GPU sanitizers
    are not available on the pool; this is native product code that runs on the CPU)"""
    if os.environ.get("SMOQY_MEMBER_LIB"):
        pytest.skip("already the sanitizer child run")
    so = str(tmp_path / "libsmoqy_member_asan.so")
    csrc = os.path.join(ROOT, "smoqyelphqmc.jl_amd", "csrc")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-Wall", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared", "-o", so,
                        os.path.join(csrc, "member.cpp"), "-pthread", "-lrt"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), "libasan.so not found next to gcc"
    env = dict(os.environ, SMOQY_MEMBER_LIB=so, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__)], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert " passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
