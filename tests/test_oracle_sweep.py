"""oracle/sweep.py — the CPU restatement of the sweep bench.py times (test infrastructure): its FFT leapfrog against the dense
oracle/efa.py, its vectorised field refresh against orc.fields_from_phonons, and the sweep itself (deterministic, reversible, a small
energy error).  No GPU."""
import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from oracle import efa
from oracle import oracle as orc
from oracle import sweep as osw

lat = sq.lattice


@pytest.mark.parametrize("frozen", [False, True])
def test_fft_leapfrog_equals_the_dense_oracle(frozen):
    g = np.random.default_rng(5)
    Nph, Lt = 7, 12
    q, m = efa.harmonic_tables(np.full(Nph, 0.9), np.full(Nph, 1.3), 0.05, Lt)
    if frozen:
        q[-1, :] = np.inf
        m[-1, :] = np.inf
    x, p, f = g.standard_normal((Nph, Lt)), g.standard_normal((Nph, Lt)), g.standard_normal((Nph, Lt))
    if frozen:
        x[-1] = 0.0
        p[-1] = 0.0
    xa, pa = efa.evolve_eom(x, p, 0.37, q, m, force=f, kick=0.11)
    xb, pb = osw.evolve_eom_fft(x, p, 0.37, q, m, force=f, kick=0.11)
    np.testing.assert_allclose(xb, xa, atol=1e-13)
    np.testing.assert_allclose(pb, pa, atol=1e-13)
    R = g.standard_normal((Nph, Lt))
    p0, K0 = efa.initialize_momentum(R, m)
    p1, K1 = osw.initialize_momentum_fft(R, m)
    np.testing.assert_allclose(p1, p0, atol=1e-13)
    assert abs(K1 - K0) < 1e-12 * K0
    Kd = efa.kinetic_energy(p, m)
    assert abs(osw.kinetic_energy_fft(p, m) - Kd) < 1e-12 * abs(Kd)
    Sd = efa.bosonic_action(x, q, m)
    assert abs(osw.bosonic_action_fft(x, q, m) - Sd) < 1e-12 * abs(Sd)


@pytest.mark.parametrize("name", ["holstein_honeycomb_L4_Ltau40", "ossh_square_L12_Ltau100", "bssh_chain_L256_Ltau200_alpha0p2"])
def test_vectorised_field_refresh_equals_the_loop(name):
    w = osw.OracleWalker(name, walker=3)
    w.x[: w.Nph] += 0.1 * np.random.default_rng(2).standard_normal((w.Nph, w.Lt))
    V, t = w.fields_from_phonons()
    w.fc.x = w.x
    Vo, to = orc.fields_from_phonons(w.fc, w.V0, w.t0, w.perm)
    np.testing.assert_allclose(V, Vo, rtol=0, atol=1e-14)
    np.testing.assert_allclose(t, to, rtol=0, atol=1e-14)


def test_sweep_is_deterministic_reversible_and_nearly_energy_conserving():
    a = osw.OracleWalker("holstein_honeycomb_L4_Ltau40", walker=1)
    b = osw.OracleWalker("holstein_honeycomb_L4_Ltau40", walker=1)
    x0 = a.x.copy()
    ra, rb = a.sweep(), b.sweep()
    assert ra["iters"] == rb["iters"] and ra["dH"] == rb["dH"]
    assert len(ra["iters"]) == 27 and a.solves == 27 and a.iters_sum == sum(ra["iters"])
    np.testing.assert_allclose(a.x, x0, rtol=0, atol=1e-15)   # every move is rejected: x restored (x + dx - dx: to rounding, as on the device)
    assert abs(ra["dH"]) < 0.05                        # 24 leapfrog steps of dt = pi/48 (1e-3 .. 1e-2 on this lattice)
    # the force solves get shorter along the trajectory (the fields smooth out): what makes this sweep a different workload from
    # 24 solves on the i.i.d. start fields, and why the CPU baseline has to follow the trajectory
    f = ra["iters"][2:26]
    assert f[-1] < f[0]
