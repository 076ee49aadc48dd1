"""Device-side update! from the phonon fields: smoqy_update_from_phonons_all
against the oracle chain fields_from_phonons -> update_fields / update_lambda, for general couplings
(α … α₄, several couplings per site / bond) and for the three synthetic model kinds, several walkers per
handle; plus smoqy_force_store_v == smoqy_force_v into zeros."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle import oracle as orc
from test_oracle_phonon_fields import general_couplings

pytestmark = pytest.mark.gpu
lat = sq.lattice


def check_fields(h, nw, fcs, V0, t0, nt, perm, is_sym, Lt, N):
    Nh = nt.shape[1]
    for w in range(nw):
        fc = fcs[w]
        V, t = orc.fields_from_phonons(fc, V0, t0, perm)
        expV, ch, sh = orc.update_fields(V, t, perm, fc.dtau, is_sym)
        Lam = orc.update_lambda(Lt, N, fc.x, fc.dtau, fc.h_c2p, fc.h_c2s, fc.h_alpha, fc.h_alpha3, fc.h_phsym)
        e, c, s = np.zeros((Lt, N), order="F"), np.zeros((Lt, Nh), order="F"), np.zeros((Lt, Nh), order="F")
        h.call("smoqy_get_fields", w, L.ptr(e), L.ptr(c), L.ptr(s))
        lam = np.zeros((Lt, N), order="F")
        h.call("smoqy_lambda_get", w, L.ptr(lam))
        for got, want in ((e, expV), (c, ch), (s, sh), (lam, Lam)):
            assert np.abs(got - want).max() < 4e-15 * np.abs(want).max()


@pytest.mark.parametrize("is_sym", [True, False])
def test_update_from_phonons_general_couplings(is_sym):
    m = lat.ossh_square(4, 6)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    Lt, N, Nh, nw = 6, m.fpi.N, nt.shape[1], 3
    fcs = [general_couplings(nt, N, Lt, 0.05, 11) for _ in range(nw)]
    g = np.random.default_rng(4)
    for w in range(1, nw):  # same couplings, different fields
        fcs[w].x[...] = 0.6 * g.standard_normal(fcs[w].x.shape)
        fcs[w].x[-1] = 0.0
    V0, t0 = 0.3 * g.standard_normal(N), 1.0 + 0.2 * g.standard_normal(Nh)
    t0[3] = -0.7  # a negative hopping: sign(t) sinh|Δτ t| (FermionDetMatrix.jl:231)
    h = L.Handle(Lt, N, nt, colors, is_sym, nw, 1)
    s, keep = L.couplings_struct(fcs[0])
    h.call("smoqy_force_set_couplings", C.byref(s))
    h.call("smoqy_set_bare_model", L.ptr(V0), L.ptr(t0), L.ptr(perm))
    xs = np.ascontiguousarray(np.stack([fc.x.T for fc in fcs]))
    h.call("smoqy_update_from_phonons_all", L.ptr(xs))
    check_fields(h, nw, fcs, V0, t0, nt, perm, is_sym, Lt, N)
    # the operator built from those fields acts like the oracle's
    fc = fcs[1]
    V, t = orc.fields_from_phonons(fc, V0, t0, perm)
    o = orc.OracleFDM(nt, *orc.update_fields(V, t, perm, fc.dtau, is_sym), is_sym)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    h.call("smoqy_matvec_v", L.OP_MTM, b, a)
    want = o.mul_MtM(v[:, :, 1])
    assert np.abs(h.vec_download(b)[:, :, 1] - want).max() < 1e-13 * np.abs(want).max()
    # second call after the fields moved (hoppings must follow)
    for fc in fcs:
        fc.x[:-1] += 0.05
    xs = np.ascontiguousarray(np.stack([fc.x.T for fc in fcs]))
    h.call("smoqy_update_from_phonons_all", L.ptr(xs))
    check_fields(h, nw, fcs, V0, t0, nt, perm, is_sym, Lt, N)


def test_update_from_phonons_needs_setup():
    m = lat.holstein_honeycomb(3, 10)
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    h = L.Handle(10, 18, nt, colors, True, 1, 1)
    with pytest.raises(L.SmoqyError):
        h.call("smoqy_update_from_phonons_all", L.ptr(np.zeros((10, 18))))


@pytest.mark.parametrize("name", ["holstein_honeycomb_L4_Ltau40", "ossh_square_L12_Ltau100", "bssh_chain_L256_Ltau200"])
def test_walker_batch_device_update_equals_host_update(name):
    """WalkerBatch with the device-side update! against the host path (V, t formed with numpy and sent
    through smoqy_update_from_path_integral_all / smoqy_lambda_update_all), after a field move."""
    nw = 2
    a = WalkerBatch(name, nwalkers=nw, device_update=True)
    b = WalkerBatch(name, nwalkers=nw, device_update=False)
    dx = 0.01 * np.random.default_rng(1).standard_normal(a.xs.shape)
    a.drift_by(dx)
    b.drift_by(dx)
    Lt, N, Nh = a.Lt, a.N, a.Nh
    for w in range(nw):
        fa = [np.zeros((Lt, n), order="F") for n in (N, Nh, Nh)]
        fb = [np.zeros((Lt, n), order="F") for n in (N, Nh, Nh)]
        a.h.call("smoqy_get_fields", w, *[L.ptr(f) for f in fa])
        b.h.call("smoqy_get_fields", w, *[L.ptr(f) for f in fb])
        for x, y in zip(fa, fb):
            assert np.abs(x - y).max() < 4e-15 * np.abs(y).max()
        la, lb = np.zeros((Lt, N), order="F"), np.zeros((Lt, N), order="F")
        a.h.call("smoqy_lambda_get", w, L.ptr(la))
        b.h.call("smoqy_lambda_get", w, L.ptr(lb))
        assert np.abs(la - lb).max() < 4e-15 * np.abs(lb).max()
    # same solve, same force on both
    for wb in (a, b):
        wb.sample_pseudofermion_fields()
    ra = a.calculate_fermionic_action(1e-10)
    rb = b.calculate_fermionic_action(1e-10)
    assert np.array_equal(ra[1], rb[1])
    np.testing.assert_allclose(ra[0], rb[0], rtol=1e-9)
    fa_ = a.fermionic_force().copy()
    out = np.zeros_like(b.dSdx)
    b.h.call("smoqy_force_v", b.u, L.ptr(out))
    np.testing.assert_allclose(fa_, out, rtol=0, atol=1e-9 * np.abs(out).max())


@pytest.mark.parametrize("name", ["holstein_honeycomb_L4_Ltau40", "bssh_chain_L256_Ltau200"])
def test_fused_pff_step_equals_the_separate_calls(name):
    """smoqy_pff_step_v (update! + update_preconditioner! + action + force in one call) against the same sequence made of the
    separate entry points, on identical inputs and random vectors: bit-identical."""
    nw = 2
    a = WalkerBatch(name, nwalkers=nw)
    b = WalkerBatch(name, nwalkers=nw)
    for wb in (a, b):
        wb.sample_pseudofermion_fields()
    dx = 0.01 * np.random.default_rng(1).standard_normal(a.xs.shape)
    # separate calls
    a.drift_by(dx)
    sfa, ita, epa = a.calculate_fermionic_action(1e-8)
    fa = a.fermionic_force().copy()
    # fused
    np.add(b.xs, dx, out=b.xs)
    sfb, itb, epb, fb = b.pff_step(1e-8, moved=True, want_force=True)
    assert np.array_equal(ita, itb) and np.array_equal(sfa, sfb) and np.array_equal(epa, epb)
    assert np.array_equal(fa, fb)
    assert np.array_equal(a.h.vec_download(a.u), b.h.vec_download(b.u))
    # without a field move and without the force
    sfa2, ita2, _ = a.calculate_fermionic_action(1e-10)
    sfb2, itb2, _ = b.pff_step(1e-10, moved=False, want_force=False)
    assert np.array_equal(ita2, itb2) and np.array_equal(sfa2, sfb2)
