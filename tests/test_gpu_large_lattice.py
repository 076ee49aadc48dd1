"""Lattices whose time slices do not fit in LDS (N > 2556 sites): the generic kernels stage their slices in a global scratch
area instead.  Every operation of the path against the CPU oracle at N = 2592 (honeycomb L = 36) and N = 4096 (chain)."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("kind,is_sym", [("honeycomb", True), ("honeycomb", False), ("chain", True)])
def test_large_lattice_against_oracle(kind, is_sym):
    m = lat.holstein_honeycomb(36, 6) if kind == "honeycomb" else lat.bssh_chain(4096, 4)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N = m.fpi.Ltau, m.fpi.N
    assert N > 2556
    nsys = 2
    h = L.Handle(Lt, N, nt, colors, is_sym, 1, nsys)
    h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    g = np.random.default_rng(2)
    v = np.asfortranarray(g.standard_normal((Lt, N, nsys)) + 1j * g.standard_normal((Lt, N, nsys)))
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    for op, fn in ((L.OP_M, o.mul_M), (L.OP_MT, o.mul_Mt), (L.OP_MTM, o.mul_MtM), (L.OP_MMT, o.mul_MMt)):
        h.call("smoqy_matvec_v", op, b, a)
        got = h.vec_download(b)
        for s in range(nsys):
            assert relerr(got[:, :, s], fn(v[:, :, s])) < 1e-13, (kind, is_sym, op, s)
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_set_tau_chunk", 2)
    # checkerboard_lmul! on the full colour range, in place
    h.vec_upload(b, v)
    h.call("smoqy_checkerboard_v", b, 0, 0, 0, colors.shape[1])
    assert relerr(h.vec_download(b)[:, :, 1], o.checkerboard(v[:, :, 1])) < 1e-13
    # KPM preconditioner and the preconditioned CG
    rv = np.random.default_rng(3).standard_normal(N)
    P = orc.OracleKPM(o)
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, nsys)
    assert relerr(out[:, :, 0], P.apply(v[:, :, 0])) < 1e-10
    x = np.zeros_like(v)
    iters, eps = np.zeros(nsys, dtype=np.int32), np.zeros(nsys)
    h.call("smoqy_cg_solve", L.ptr(x), L.ptr(v), 1, 0, nsys, C.c_double(1e-10), 20000, 1, L.ptr(iters), L.ptr(eps))
    xo, ito, _ = o.cg_solve(v[:, :, 0], precond=P, tol=1e-10, maxiter=20000)
    assert abs(int(iters[0]) - ito) <= 2 and eps.max() < 1e-10
    assert relerr(o.mul_MtM(x[:, :, 0]), v[:, :, 0]) < 1e-8  # max-norm of the true residual; the stop test is on the 2-norm (1e-10)


def test_large_lattice_force():
    m = lat.holstein_honeycomb(36, 4)
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-10)
    fc = m.force_couplings(fdm.checkerboard_perm)
    sq.set_force_couplings(fdm, fc)
    Lt, N = fdm.Lt, fdm.N
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    o = orc.OracleFDM(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    g = np.random.default_rng(5)
    u = np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))
    v = np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))
    got = np.zeros((fc.x.shape[0], Lt), order="F")
    sq.mul_nuRe_dMdx(got, 1.3, u, v, fdm)
    want = orc.mul_dMdx(o, orc.OracleElph(fc), fdm._colors, 1.3, u, v)
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()


@pytest.mark.parametrize("is_sym", [True, False])
def test_widest_owner_computes_workgroups(is_sym):
    """Honeycomb L = 32: N = 2048 sites, 1024 padded bonds per colour — the widest workgroup of the register-resident / owner-computes
    kernels (1024 lanes).  The complex (non-split) owner-computes Chebyshev kernels then need 4 x 1024 x 16 bytes of LDS images plus the
    coefficient tables (just past 64 KB; configure_kpm_kernels states the limit explicitly): the Sym real-vector ldiv!
    (half spectrum: not split, KPMPreconditioner.jl:288-352) and the Asym apply (KPMPreconditioner.jl:488-550) against the oracle."""
    m = lat.holstein_honeycomb(32, 8)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N = m.fpi.Ltau, m.fpi.N
    assert N == 2048
    nsys = 2
    h = L.Handle(Lt, N, nt, colors, is_sym, 1, nsys)
    h.call("smoqy_update_from_path_integral", 0, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, is_sym)
    o = orc.OracleFDM(nt, expV, ch, sh, is_sym)
    g = np.random.default_rng(7)
    v = np.asfortranarray(g.standard_normal((Lt, N, nsys)) + 1j * g.standard_normal((Lt, N, nsys)))
    rv = g.standard_normal(N)
    P = orc.OracleKPM(o)
    P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    out = np.zeros_like(v)
    h.call("smoqy_precond_apply", L.ptr(out), L.ptr(v), 0, nsys)
    for s in range(nsys):
        assert relerr(out[:, :, s], P.apply(v[:, :, s])) < 1e-10, s
    if is_sym:
        vr = np.asfortranarray(g.standard_normal((Lt, N, nsys)))
        outr = np.zeros_like(vr)
        h.call("smoqy_precond_apply_real", L.ptr(outr), L.ptr(vr), 0, nsys)
        assert relerr(outr[:, :, 1], P.apply_real(vr[:, :, 1])) < 1e-10
    x = np.zeros_like(v)
    iters, eps = np.zeros(nsys, dtype=np.int32), np.zeros(nsys)
    h.call("smoqy_cg_solve", L.ptr(x), L.ptr(v), 1, 0, nsys, C.c_double(1e-10), 20000, 1, L.ptr(iters), L.ptr(eps))
    xo, ito, _ = o.cg_solve(v[:, :, 0], precond=P, tol=1e-10, maxiter=20000)
    assert abs(int(iters[0]) - ito) <= 1 and eps.max() < 1e-10
    assert relerr(x[:, :, 0], xo) < 1e-8
