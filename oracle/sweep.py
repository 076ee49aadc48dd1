"""One walker's synthetic sweep on the CPU oracle.  TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

The sweep ``bench.py`` times on the GPU (``smoqyelphqmc_amd.walkers.WalkerBatch._sweep`` with ``device_efa=True``), restated with the
oracle's own pieces so that the CPU baseline and the GPU figure are the SAME work on the SAME numbers:

  two global moves      Φ = Λᵀ Mᵀ R  (src/PFFCalculator.jl:56-76), x ← x + δπ, update!(fdm, fpi) + update_Λ!, one action solve at tol
                        (src/PFFCalculator.jl:79-116), x restored  (the shape of src/reflection_update.jl:69-114 / src/swap_update.jl)
  hmc_update!           src/EFAPFFHMCUpdater.jl:102-276: Φ sampled (:133), S_b (:136), momenta (:142), evolve(Δt/2) (:148-152),
                        Nt × { ∂S_f/∂x behind a solve at √tol (:160-165, src/PFFCalculator.jl:119-157), p −= Δt ∂S/∂x (:196),
                        evolve(Δt, Δt/2 at the last step), update! (:200-205) }, final action at tol (:217), ΔH (:234-250),
                        the move always rejected (x ← x0, :263-275), as the bench does.

Every solve is preceded by update_preconditioner! (src/FermionDetMatrix.jl:259) with a fresh start vector (src/KPMPreconditioner.jl:634).
The walker's generator is PCG64(SEED0 + 7919·walker + 1) and is asked for the same arrays in the same order as WalkerBatch asks
(``WalkerBatch._fill_draws``), so device and oracle run the same Markov-chain step: iteration counts, actions and ΔH are comparable
number for number (tests/test_gpu_sweep_parity.py).

The EFA leapfrog is oracle/efa.py's algorithm (PARITY UNPINNED, see its header) with numpy's FFT in place of the dense DFT matrices —
``evolve_eom_fft`` is checked against ``efa.evolve_eom`` in tests/test_oracle_sweep.py; the dense form would put an O(Lτ²) matrix
product per step into the timed CPU baseline that the reference (FFTW) does not have.
"""
from __future__ import annotations

import time

import numpy as np

import smoqyelphqmc_amd as sq
from oracle import efa
from oracle import oracle as orc

lat = sq.lattice


def _live(m):
    return np.isfinite(m) & (m > 0)


def evolve_eom_fft(x, p, dt, q, m, force=None, kick=0.0):
    """efa.evolve_eom with FFTs along τ (same rotation per (phonon, ω), same treatment of frozen modes)."""
    x, p = np.array(x, dtype=float), np.array(p, dtype=float)
    if force is not None:
        p = p - kick * np.asarray(force)
    Lt = x.shape[1]
    xt, pt = np.fft.fft(x, axis=1), np.fft.fft(p, axis=1)  # the 1/√Lτ of the unitary transform cancels between forward and inverse
    live = _live(m)
    mm = np.where(live, m, 1.0)
    w = np.sqrt(np.where(live, q / mm, 0.0))
    c, s = np.cos(w * dt), np.sin(w * dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = np.where(w > 0, s / (mm * w), dt / mm)
    xn = np.where(live, c * xt + f1 * pt, xt)
    pn = np.where(live, c * pt - mm * w * s * xt, pt)
    xo, po = np.fft.ifft(xn, axis=1).real, np.fft.ifft(pn, axis=1).real
    frozen = ~live.any(axis=1)
    xo[frozen], po[frozen] = x[frozen], p[frozen]
    return xo, po


def initialize_momentum_fft(R, m):
    """efa.initialize_momentum with FFTs: p = F⁻¹ √m F R, K = ½ Σ |p̃|²/m."""
    live = _live(m)
    rt = np.fft.fft(np.asarray(R, dtype=float), axis=1)
    pt = np.where(live, np.sqrt(np.where(live, m, 0.0)), 0.0) * rt
    p = np.fft.ifft(pt, axis=1).real
    return p, kinetic_energy_fft(p, m)


def kinetic_energy_fft(p, m):
    pt = np.fft.fft(np.asarray(p, dtype=float), axis=1) / np.sqrt(np.shape(p)[1])
    live = _live(m)
    return float(0.5 * np.sum(np.where(live, np.abs(pt) ** 2 / np.where(live, m, 1.0), 0.0)))


def bosonic_action_fft(x, q, m):
    xt = np.fft.fft(np.asarray(x, dtype=float), axis=1) / np.sqrt(np.shape(x)[1])
    live = _live(m)
    return float(0.5 * np.sum(np.where(live, np.where(live, q, 0.0) * np.abs(xt) ** 2, 0.0)))


class OracleWalker:
    """One walker of ``workload`` with everything the sweep touches held on the host."""

    def __init__(self, workload, walker=0, tol=1e-10, maxiter=10_000, Nt=24, drift=0.02, omega=1.0, mass=1.0):
        m = lat.CONFIGS[workload](walker=walker)
        self.model = m
        self.nt, self.perm, self.colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
        self.fc = m.force_couplings(self.perm)
        self.Lt, self.N = m.fpi.Ltau, m.fpi.N
        self.Nph = m.elph.x.shape[0]
        self.dtau = m.fpi.dtau
        self.tol, self.tol_force, self.maxiter, self.Nt, self.drift = tol, float(np.sqrt(tol)), maxiter, Nt, drift
        self.elph = orc.OracleElph(self.fc)           # holds x as (Nph_force, Lτ); the C struct points at elph.x
        self.x = self.elph.x
        self.V0, self.t0 = m.bare_model()
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, self.perm, self.dtau, True)
        self.fdm = orc.OracleFDM(self.nt, expV, ch, sh, True)   # the C side reads fdm.expV / cosh / sinh in place
        self.P = orc.OracleKPM(self.fdm)
        self.Lam = np.ones((self.Lt, self.N), order="F")
        self.rng = np.random.Generator(np.random.PCG64(lat.SEED0 + 7919 * walker + 1))
        fm = np.asarray(self.fc.finite_mass, dtype=bool)
        row = self.dtau * mass * (omega**2 + 4.0 / self.dtau**2 * np.sin(np.pi * np.arange(self.Lt) / self.Lt) ** 2)
        self.q = np.where(fm[:, None], row[None, :], np.inf)     # WalkerBatch.efa_setup
        self.m = self.q.copy()
        # vectorised fields_from_phonons (index arrays built once)
        c = self.fc
        self._h_p, self._h_s = np.asarray(c.h_c2p, dtype=np.int64) - 1, np.asarray(c.h_c2s, dtype=np.int64) - 1
        s_c2p = np.asarray(c.s_c2p, dtype=np.int64).reshape(2, -1)
        self._s_p0, self._s_p1 = s_c2p[0] - 1, s_c2p[1] - 1
        self._s_h = np.asarray(self.perm, dtype=np.int64)[np.asarray(c.s_bond, dtype=np.int64) - 1] - 1
        self.solves = self.iters_sum = 0
        self.refresh_fields()

    # ---- update!(fdm, fpi), update_Λ! from the current x ---------------------------------------------------------------
    def fields_from_phonons(self):
        """orc.fields_from_phonons without the per-coupling Python loop (checked against it in tests/test_oracle_sweep.py)."""
        c, x = self.fc, self.x
        V = np.repeat(np.asarray(self.V0, dtype=float)[:, None], self.Lt, axis=1)
        t = np.repeat(np.asarray(self.t0, dtype=float)[:, None], self.Lt, axis=1)
        def poly(z, a1, a2, a3, a4):  # α z + α₂ z² + α₃ z³ + α₄ z⁴, terms with all-zero couplings skipped
            out = np.asarray(a1, dtype=float)[:, None] * z
            for k, a in ((2, a2), (3, a3), (4, a4)):
                a = np.asarray(a, dtype=float)
                if np.any(a != 0.0):
                    out += a[:, None] * z**k
            return out

        if len(self._h_p):
            np.add.at(V, self._h_s, poly(x[self._h_p], c.h_alpha, c.h_alpha2, c.h_alpha3, c.h_alpha4))
        if len(self._s_h):
            np.subtract.at(t, self._s_h, poly(x[self._s_p1] - x[self._s_p0], c.s_alpha, c.s_alpha2, c.s_alpha3, c.s_alpha4))
        return np.asfortranarray(V), np.asfortranarray(t)

    def refresh_fields(self):
        V, t = self.fields_from_phonons()
        expV, ch, sh = orc.update_fields(V, t, self.perm, self.dtau, True)   # src/FermionDetMatrix.jl:208-236
        self.fdm.expV[...], self.fdm.cosh[...], self.fdm.sinh[...] = expV, ch, sh
        c = self.fc
        self.Lam = orc.update_lambda(self.Lt, self.N, self.x, self.dtau, c.h_c2p, c.h_c2s, c.h_alpha, c.h_alpha3, c.h_phsym)  # src/holstein_shift_matrix.jl:2-44

    # ---- PFFCalculator ------------------------------------------------------------------------------------------------
    def _cn(self):
        """randn!(rng, Φ) for ComplexF64: (re, im) pairs in memory order, each of variance 1/2."""
        flat = self.rng.standard_normal(2 * self.Lt * self.N) * np.sqrt(0.5)
        return np.asfortranarray(flat.view(np.complex128).reshape(self.Lt, self.N, order="F"))

    def sample_pseudofermion_fields(self):
        R = self._cn()
        self.phi = orc.lambda_apply(self.Lam, self.fdm.mul_Mt(R), "mulT")      # src/PFFCalculator.jl:67-73
        return float(np.vdot(R, R).real)

    def action(self, tol, rv):
        """calculate_fermionic_action! (src/PFFCalculator.jl:79-116): returns (S_f, iters, eps), leaves Ψ in self.psi."""
        self.P.update(rv)
        b = orc.lambda_apply(self.Lam, self.phi, "ldivT")
        xs, it, eps = self.fdm.cg_solve(b, precond=self.P, tol=tol, maxiter=self.maxiter)
        self.psi = orc.lambda_apply(self.Lam, xs, "ldiv")
        self.solves += 1
        self.iters_sum += it
        return float(np.vdot(self.phi, self.psi).real), it, eps

    def force(self):
        """The tail of calculate_derivative_fermionic_action! (src/PFFCalculator.jl:146-155) from the Ψ of the last solve."""
        LPsi = orc.lambda_apply(self.Lam, self.psi, "mul")
        APsi = self.fdm.mul_M(LPsi)
        dS = orc.mul_dMdx(self.fdm, self.elph, self.colors, -2.0, APsi, LPsi)
        orc.mul_dLdx(self.elph, self.Lam, -2.0, self.fdm.mul_Mt(APsi), self.psi, dS)
        return dS

    # ---- the sweep ----------------------------------------------------------------------------------------------------
    def sweep(self):
        """Returns a dict with the per-solve iteration counts (in order), the last action and ΔH of the trajectory."""
        g, iters = self.rng, []
        for _ in range(2):
            self.sample_pseudofermion_fields()
            dx = g.standard_normal((self.Lt, self.Nph)) * self.drift     # WalkerBatch draws (Lτ, Nph) per walker
            rv = g.standard_normal(self.N)
            self.x[: self.Nph] += dx.T
            self.refresh_fields()
            iters.append(self.action(self.tol, rv)[1])
            self.x[: self.Nph] -= dx.T
            self.refresh_fields()
        # hmc_update!
        x0 = self.x.copy()
        dt = np.pi / (2 * self.Nt)                                        # tutorials/holstein_honeycomb.jl:542
        sf0 = self.sample_pseudofermion_fields()
        R = g.standard_normal((self.Lt, self.x.shape[0])).T              # (Nph_force, Lτ); frozen modes consume deviates like the device does
        rvs = [g.standard_normal(self.N) for _ in range(self.Nt)]
        rv_last = g.standard_normal(self.N)
        p, K0 = initialize_momentum_fft(R, self.m)
        Sb0 = bosonic_action_fft(self.x, self.q, self.m)
        xn, p = evolve_eom_fft(self.x, p, dt / 2, self.q, self.m)
        self.x[...] = xn
        self.refresh_fields()
        for t in range(self.Nt):
            iters.append(self.action(self.tol_force, rvs[t])[1])
            dS = self.force()
            xn, p = evolve_eom_fft(self.x, p, dt / 2 if t == self.Nt - 1 else dt, self.q, self.m, force=dS, kick=dt)
            self.x[...] = xn
            self.refresh_fields()
        sf1, it, _ = self.action(self.tol, rv_last)
        iters.append(it)
        K1, Sb1 = kinetic_energy_fft(p, self.m), bosonic_action_fft(self.x, self.q, self.m)
        dH = (sf1 + Sb1 + K1) - (sf0 + Sb0 + K0)
        self.x[...] = x0                                                  # rejected, :263-275
        self.refresh_fields()
        return {"iters": iters, "action": sf1, "dH": dH}


def timed_sweeps(workload, walker, nsweeps, tol=1e-10, Nt=24):
    """What bench.py's cpu_baseline leg runs per core: `nsweeps` whole sweeps, nothing extrapolated."""
    w = OracleWalker(workload, walker=walker, tol=tol, Nt=Nt)
    t0 = time.perf_counter()
    its = []
    for _ in range(nsweeps):
        its += w.sweep()["iters"]
    return time.perf_counter() - t0, its, w
