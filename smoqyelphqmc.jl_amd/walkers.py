"""Lock-step batch of independent Monte Carlo walkers on one GPU, and the synthetic "sweep"
that ``bench.py`` times.

The reference runs one walker per MPI rank with no state exchange
(tutorials/holstein_honeycomb_mpi.jl:60-72); here every rank (= GPU) owns ``nwalkers`` walkers
whose pseudofermion-action solves advance in lock step through one ``smoqy_ctx`` so that all
kernels are launched once for the whole batch (SURVEY.md §7 hard part 1).

A sweep follows tutorials/holstein_honeycomb.jl:611-684: reflection update + swap update +
one HMC update with ``Nt`` force evaluations, i.e. ``1 + 1 + (Nt + 1)`` CG solves
(27 at ``Nt = 24``).  Everything the hot path owns runs on the device exactly as in the
reference call stack:

  sample_pseudofermion_fields!   Φ = Λᵀ Mᵀ R                        src/PFFCalculator.jl:56-76
  calculate_fermionic_action!    Ψ = Λ⁻¹ (MᵀM)⁻¹ Λ⁻ᵀ Φ, S = Φ·Ψ     src/PFFCalculator.jl:79-116
  force (∂S_f/∂x)                ΛΨ, MΛΨ, ∂M/∂x, MᵀMΛΨ, ∂Λ/∂x        src/PFFCalculator.jl:146-155
  update!(fdm, fpi)              exp/cosh/sinh refresh               src/FermionDetMatrix.jl:208-236
  update_preconditioner!         B̄ means, Lanczos, KPM coefficients  src/KPMPreconditioner.jl:554-597

What is NOT part of the hot path (SmoQyDQMC's EFA leapfrog and bosonic action) is replaced by a synthetic drift of the phonon field
``x ← x + δ·π`` with a fixed random "momentum" π, so successive solves see slowly moving
fields like an HMC trajectory does.  Random numbers are drawn on the host, one generator per
walker (the reference's rng stays on the host).
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from . import lattice as lat


@dataclass
class SweepStats:
    solves: int = 0
    iters_sum: int = 0
    action: float = 0.0


class WalkerBatch:
    # optional threading.Semaphore shared by the batches of one GPU: bounds how many batches are inside the CG at once
    # (the other batches' host work and transfers still overlap with them)
    solve_gate = None

    def __init__(self, workload: str, nwalkers: int = 1, walker0: int = 0, is_sym: bool = True, device: int = -1, tol: float = 1e-10, maxiter: int = 10_000, Nt: int = 24,
                 drift: float = 0.02, check_every: int | None = None, tau_chunk: int | None = None, smooth: bool = False, host_threads: int = 8, device_update: bool = True, measure_nrv: int = 0,
                 device_efa: bool = False, omega: float = 1.0, mass: float = 1.0, cg_split: int | None = None, tfft_in_place: bool | None = None,
                 prefetch_randoms: bool = False):
        self.models = [lat.CONFIGS[workload](walker=walker0 + w, smooth=smooth) for w in range(nwalkers)]
        m0 = self.models[0]
        self.workload = workload
        self.nt, self.perm, self.colors = lat.checkerboard_decomposition(m0.fpi.neighbor_table)
        self.Lt, self.N, self.nw = m0.fpi.Ltau, m0.fpi.N, nwalkers
        self.Nh, self.Nph = m0.fpi.t.shape[0], m0.elph.x.shape[0]
        self.dtau = m0.fpi.dtau
        self.tol, self.tol_force, self.maxiter, self.Nt, self.drift = tol, float(np.sqrt(tol)), maxiter, Nt, drift  # tutorials/holstein_honeycomb.jl:591
        # stacked host storage: walker w's (Nph x Ltau) / (N x Ltau) / (Nh x Ltau) column-major arrays
        # are the transposed views xs[w].T etc., so the whole batch crosses the C ABI in one call
        self.h = L.Handle(self.Lt, self.N, self.nt, self.colors, is_sym, nwalkers, 1, device)
        self.xs = self.h.pinned_empty((nwalkers, self.Lt, self.Nph))   # page-locked: crosses the boundary on every field move
        self.Vs = self.h.pinned_empty((nwalkers, self.Lt, self.N))
        self.ts = self.h.pinned_empty((nwalkers, self.Lt, self.Nh))
        for w, m in enumerate(self.models):
            self.xs[w] = m.elph.x.T
            self.Vs[w] = m.fpi.V.T
            self.ts[w] = m.fpi.t.T
            m.elph.x, m.fpi.V, m.fpi.t = self.xs[w].T, self.Vs[w].T, self.ts[w].T
        self.hoppings_move = m0.kind != "holstein"  # Holstein: t is constant, only V follows the phonons
        if check_every:
            self.h.call("smoqy_cg_config", int(check_every))
        self.cg_split = cg_split
        if cg_split is not None:
            self.h.call("smoqy_cg_split", int(cg_split))  # 0 automatic, 1 off, 2 on: two-part pipeline of the CG loop inside the handle
        self.tfft_in_place = tfft_in_place
        if tfft_in_place is not None:
            self.h.call("smoqy_tfft_form", int(bool(tfft_in_place)))  # in-place τ-FFT: slower alone, faster when several batches share the GPU
        if tau_chunk:
            self.h.call("smoqy_set_tau_chunk", int(tau_chunk))
        self.rng = [np.random.Generator(np.random.PCG64(lat.SEED0 + 7919 * (walker0 + w) + 1)) for w in range(nwalkers)]
        self.pool = ThreadPoolExecutor(max_workers=max(1, min(host_threads, nwalkers)))
        # device-resident PFFCalculator state (src/PFFCalculator.jl:9-16): Φ, u, u′, u″
        self.phi, self.u, self.u1, self.u2 = (self.h.vec_alloc() for _ in range(4))
        self._R = self.h.pinned_empty((self.Lt, self.N, nwalkers), dtype=np.complex128, order="F")
        self._tmp = np.empty_like(self.xs)
        # force terms (src/PFFCalculator.jl:146-155): couplings go to the device once
        self.force_couplings = m0.force_couplings(self.perm)
        self._cs, self._cs_keep = L.couplings_struct(self.force_couplings)
        self.h.call("smoqy_force_set_couplings", C.byref(self._cs))
        self.Nph_force = int(self.force_couplings.x.shape[0])
        if self.Nph_force != self.Nph:  # bond-SSH: one extra infinite-mass partner mode pinned at zero
            self.xs_force = self.h.pinned_empty((nwalkers, self.Lt, self.Nph_force))
            self.xs_force[...] = 0.0
        else:
            self.xs_force = self.xs
        self.dSdx = self.h.pinned_empty((nwalkers, self.Lt, self.Nph_force))
        # device-side update! from x: the host sends x only, V / t / Λ are formed on the device
        self.device_update = device_update
        self.measure_nrv = int(measure_nrv)
        if device_update:
            V0, t0 = m0.bare_model()
            self.h.call("smoqy_set_bare_model", L.ptr(V0), L.ptr(t0), L.ptr(self.perm))
        self.stats = SweepStats()
        self.iter_log = None   # set to a list to have every solve append its per-walker iteration counts (tests)
        self.refresh_fields(first=True)
        # EFA leapfrog on the device (parity unpinned — SmoQyDQMC's accelerator is not part of the reference tree):
        # x, p and the force stay on the GPU for the whole trajectory.  Off by default: the host-side drift stays the tested default.
        self.device_efa = bool(device_efa)
        self.dH = None
        # prefetch_randoms: the random numbers of sweep n + 1 are drawn by the host-thread pool while the device runs sweep n (each walker's
        # generator is asked for the same arrays in the same order as without it, so the Markov chain is the same one); device_efa sweeps only
        self.prefetch_randoms = bool(prefetch_randoms) and self.device_efa and not self.measure_nrv
        self._draws = None          # the sweep's pre-drawn arrays while sweep() runs, else None: every draw site asks its generator
        self._draws_next = None     # (arrays, futures) being filled for the next sweep
        self._draw_sets = None
        self.h.before_close = [self._drain_draws]  # a fill task must not outlive the page-locked arrays it writes
        if self.device_efa:
            if not device_update:
                raise ValueError("device_efa needs device_update=True")
            self.efa_setup(omega, mass)

    # ---- field plumbing -------------------------------------------------------------------------
    def refresh_fields(self, first: bool = False):
        """update!(fdm, fpi) and update_Λ! for every walker from its current phonon field."""
        if self.device_update:
            if self.xs_force is not self.xs:
                self.xs_force[:, :, : self.Nph] = self.xs
            self.h.call("smoqy_update_from_phonons_all", L.ptr(self.xs_force))
            return
        if self.models[0].kind == "holstein":
            # V = α x - μ for the whole batch in one pass (what SyntheticModel.refresh_from_x does per walker)
            np.multiply(self.xs, self.models[0].alpha, out=self.Vs)
            if self.models[0].mu != 0.0:
                self.Vs -= self.models[0].mu
        else:
            for m in self.models:
                m.refresh_from_x()
        t_all = L.ptr(self.ts) if (first or self.hoppings_move) else None
        self.h.call("smoqy_update_from_path_integral_all", L.ptr(self.Vs), t_all, L.ptr(self.perm), C.c_double(self.dtau))
        hol = self.models[0].elph.holstein
        if hol is not None:
            self.h.call("smoqy_lambda_update_all", L.ptr(self.xs), self.Nph, C.c_double(self.dtau), len(hol.alpha), L.ptr(np.ascontiguousarray(hol.coupling_to_phonon, dtype=np.int64)),
                        L.ptr(np.ascontiguousarray(hol.coupling_to_site, dtype=np.int64)), L.ptr(np.ascontiguousarray(hol.alpha, dtype=np.float64)), L.ptr(np.ascontiguousarray(hol.alpha3, dtype=np.float64)),
                        L.ptr(np.ascontiguousarray(hol.ph_sym_form, dtype=np.int32)))
        else:
            if first:
                self.h.call("smoqy_lambda_update_all", None, 0, C.c_double(self.dtau), 0, None, None, None, None, None)
            if self.xs_force is not self.xs:
                self.xs_force[:, :, : self.Nph] = self.xs
            self.h.call("smoqy_force_set_phonons", L.ptr(self.xs_force))  # Holstein models share the Λ upload instead

    def _start_vectors(self):
        """randn!(rng, v) at KPMPreconditioner.jl:634, one start vector per walker."""
        d = self._draws
        if d is not None:
            d.i_rv += 1
            return d.rv[d.i_rv - 1]
        return np.ascontiguousarray(np.stack([g.standard_normal(self.N) for g in self.rng]))

    def update_preconditioner(self):
        self.h.call("smoqy_precond_update_all", L.ptr(self._start_vectors()))

    # ---- random numbers one sweep ahead (prefetch_randoms) ---------------------------------------------
    class _Draws:
        """What one device_efa sweep asks the walkers' generators for, in the order it asks."""

        def __init__(self, b):
            self.R = [b.h.pinned_empty((b.Lt, b.N, b.nw), dtype=np.complex128, order="F") for _ in range(3)]   # sample_pseudofermion_fields x 3
            self.dx = [np.empty((b.nw, b.Lt, b.Nph)) for _ in range(2)]                                        # momenta of the two global moves, times the drift
            self.rv = [np.empty((b.nw, b.N)) for _ in range(3)]                                                # start vectors of the three action solves
            self.efa_R = b.h.pinned_empty((b.nw, b.Lt, b.Nph_force))                                           # momentum refresh of hmc_update!
            self.efa_rv = b.h.pinned_empty((b.Nt, b.nw, b.N))                                                  # start vectors of the trajectory's force solves
            self.i_R = self.i_dx = self.i_rv = 0

    def _fill_draws(self, d, w):
        g = self.rng[w]

        def cn(R):  # randn!(rng, Φ) for ComplexF64, as in sample_pseudofermion_fields
            flat = R[:, :, w].reshape(-1, order="F").view(np.float64)
            g.standard_normal(out=flat)
            flat *= np.sqrt(0.5)

        for i in range(2):  # the two global moves of sweep(): Φ, momentum, start vector of the action solve
            cn(d.R[i])
            g.standard_normal(out=d.dx[i][w])
            d.dx[i][w] *= self.drift
            g.standard_normal(out=d.rv[i][w])
        cn(d.R[2])          # hmc_trajectory_device: Φ, momentum refresh, the force solves' start vectors, the final action's
        g.standard_normal(out=d.efa_R[w].reshape(-1))
        for t in range(self.Nt):
            g.standard_normal(out=d.efa_rv[t, w])
        g.standard_normal(out=d.rv[2][w])

    def _submit_draws(self):
        if self._draw_sets is None:
            self._draw_sets = [WalkerBatch._Draws(self), WalkerBatch._Draws(self)]
        d = self._draw_sets[0] if self._draws is not self._draw_sets[0] else self._draw_sets[1]
        d.i_R = d.i_dx = d.i_rv = 0
        self._draws_next = (d, [self.pool.submit(self._fill_draws, d, w) for w in range(self.nw)])

    def _drain_draws(self):
        if self._draws_next is not None:
            for f in self._draws_next[1]:
                f.result()

    def _take_draws(self):
        if self._draws_next is None:
            self._submit_draws()
        d, futures = self._draws_next
        for f in futures:
            f.result()
        self._draws_next = None
        return d

    def _randn_all(self, shape):
        """One standard-normal array per walker from that walker's generator (host threads)."""
        return list(self.pool.map(lambda g: g.standard_normal(shape), self.rng))

    # ---- PFFCalculator on the device ---------------------------------------------------------------
    def sample_pseudofermion_fields(self):
        """Φ = Λᵀ Mᵀ R with R ~ CN(0,1) drawn on the host; returns |R|² per walker."""
        d = self._draws
        if d is not None:
            R = d.R[d.i_R]
            d.i_R += 1
        else:
            R = self._R

            def fill(w):
                # randn!(rng, Φ) for ComplexF64: (re, im) pairs in memory order, each of variance 1/2
                flat = R[:, :, w].reshape(-1, order="F").view(np.float64)
                self.rng[w].standard_normal(out=flat)
                flat *= np.sqrt(0.5)

            list(self.pool.map(fill, range(self.nw)))
        self.h.vec_upload(self.phi, R)
        sf = self.h.vec_dot(self.phi, self.phi).real
        self.h.call("smoqy_matvec_v", L.OP_MT, self.phi, self.phi)          # lmul_Mt!  (:71)
        self.h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, self.phi, self.phi)  # mul_Λᵀ!  (:73)
        return sf

    def calculate_fermionic_action(self, tol, use_precond=True):
        """Returns (S_f, iters, eps) per walker; Ψ is left in ``self.u``."""
        if use_precond:
            self.update_preconditioner()                                     # ldiv! -> update_preconditioner! (FermionDetMatrix.jl:259)
        self.h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIVT, self.u, self.phi)  # Ψ = Λ⁻ᵀ Φ   (:97)
        iters = np.zeros(self.nw, dtype=np.int32)
        eps = np.zeros(self.nw)
        gate = WalkerBatch.solve_gate
        if gate is not None:
            gate.acquire()
        try:
            self.h.call("smoqy_cg_solve_v", self.u, self.u, C.c_double(tol), int(self.maxiter), int(bool(use_precond)), L.ptr(iters), L.ptr(eps))  # (:99)
        finally:
            if gate is not None:
                gate.release()
        self.h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIV, self.u, self.u)   # Ψ = Λ⁻¹ Ψ   (:107)
        sf = self.h.vec_dot(self.phi, self.u)                                 # S = Φ·Ψ     (:109)
        self.stats.solves += self.nw
        self.stats.iters_sum += int(iters.sum())
        if self.iter_log is not None:
            self.iter_log.append(iters.copy())
        return sf.real, iters, eps

    def fermionic_force(self):
        """The tail of calculate_derivative_fermionic_action! (src/PFFCalculator.jl:146-155):
        ΛΨ, AΨ = MΛΨ, -2 Re⟨AΨ|∂M/∂x|ΛΨ⟩, MᵀAΨ, -2 Re⟨MᵀAΨ|∂Λ/∂x|Ψ⟩ — on the device; the force array
        (Nph x Ltau per walker) comes back to the host, where the reference's leapfrog consumes it."""
        self.h.call("smoqy_force_store_v", self.u, L.ptr(self.dSdx))  # fill!(∂S∂x, 0) + accumulate == store (EFAPFFHMCUpdater.jl:160-165)
        return self.dSdx

    def pff_step(self, tol, moved: bool, want_force: bool, use_precond: bool = True):
        """One HMC force evaluation in ONE library call (smoqy_pff_step_v): [update! from the moved x], update_preconditioner!,
        Ψ = Λ⁻¹ (MᵀM)⁻¹ Λ⁻ᵀ Φ, S_f = Φ·Ψ and, if wanted, ∂S_f/∂x — calculate_derivative_fermionic_action!
        (src/PFFCalculator.jl:119-157) behind src/EFAPFFHMCUpdater.jl:200-205.  Returns (S_f, iters, eps[, force])."""
        if not self.device_update:
            raise RuntimeError("pff_step needs the device-side update! (device_update=True)")
        if moved and self.xs_force is not self.xs:
            self.xs_force[:, :, : self.Nph] = self.xs
        rv = self._start_vectors() if use_precond else None  # randn!(rng, v), KPMPreconditioner.jl:634
        sf = np.zeros(self.nw)
        iters = np.zeros(self.nw, dtype=np.int32)
        eps = np.zeros(self.nw)
        gate = WalkerBatch.solve_gate
        if gate is not None:
            gate.acquire()
        try:
            self.h.call("smoqy_pff_step_v", self.phi, self.u, L.ptr(self.xs_force) if moved else None, None if rv is None else L.ptr(rv), C.c_double(tol), int(self.maxiter),
                        int(bool(use_precond)), L.ptr(sf), L.ptr(iters), L.ptr(eps), L.ptr(self.dSdx) if want_force else None)
        finally:
            if gate is not None:
                gate.release()
        self.stats.solves += self.nw
        self.stats.iters_sum += int(iters.sum())
        if self.iter_log is not None:
            self.iter_log.append(iters.copy())
        return (sf, iters, eps, self.dSdx) if want_force else (sf, iters, eps)

    # ---- measurements (GreensEstimator) -----------------------------------------------
    def measure_greens(self, Nrv: int = 10, orbitals=(1, 1), tol=None):
        """update_greens_estimator! + measure_GΔ0! (src/Measurements/GreensEstimator.jl:125-233) for every walker of the batch:
        nwalkers x Nrv right-hand sides advance in ONE batched CG on a follower handle; returns G(Δ,0) per walker
        (shape (nwalkers, Lτ+1, L...)) and the iteration counts."""
        meta = self.models[0].meta
        if "L" not in meta:
            raise ValueError("no lattice geometry recorded for this workload")
        Lc = int(meta["L"])
        n_orb = 2 if self.models[0].name.startswith("holstein_honeycomb") else 1
        Ls = (Lc, Lc) if n_orb * Lc * Lc == self.N else (Lc,)
        if getattr(self, "_ge", None) is None or self._ge[1] != Nrv:
            hg = L.Handle(self.Lt, self.N, self.nt, self.colors, True, self.nw, Nrv, self.h.device)
            if self.cg_split is not None:
                hg.call("smoqy_cg_split", int(self.cg_split))  # the measurement handle follows the batch's choice (and frees its part stream when that is 1)
            if self.tfft_in_place is not None:
                hg.call("smoqy_tfft_form", int(bool(self.tfft_in_place)))
            hg.call("smoqy_ge_config", n_orb, len(Ls), L.ptr(np.asarray(Ls, dtype=np.int64)))
            self._ge = (hg, Nrv, tuple(hg.vec_alloc() for _ in range(3)), hg.pinned_empty((self.Lt, self.N, self.nw * Nrv), dtype=np.complex128, order="F"))
        hg, _, (r, gr, mtr), R = self._ge
        for w in range(self.nw):
            hg.call("smoqy_copy_fields", w, self.h._h, w)

        def fill(w):  # randn!(rng, R); R ./= abs.(R)  (:141-142)
            blk = R[:, :, w * Nrv : (w + 1) * Nrv]
            flat = blk.reshape(-1, order="F").view(np.float64)
            self.rng[w].standard_normal(out=flat)
            np.divide(blk, np.abs(blk), out=blk)

        list(self.pool.map(fill, range(self.nw)))
        rv = np.ascontiguousarray(np.stack([g.standard_normal(self.N) for g in self.rng]))
        hg.call("smoqy_precond_update_all", L.ptr(rv))                                  # :150
        hg.vec_upload(r, R)
        hg.call("smoqy_matvec_v", L.OP_MT, mtr, r)                                      # :157
        iters = np.zeros(self.nw * Nrv, dtype=np.int32)
        eps = np.zeros(self.nw * Nrv)
        hg.call("smoqy_cg_solve_v", gr, mtr, C.c_double(self.tol if tol is None else tol), int(self.maxiter), 1, L.ptr(iters), L.ptr(eps))  # :159-165
        out = np.zeros((self.nw, *reversed(Ls), self.Lt + 1), dtype=np.complex128)       # C order == (Lτ+1) x L... x nw column-major
        hg.call("smoqy_ge_measure_GD0", gr, r, int(orbitals[0]), int(orbitals[1]), L.ptr(out))
        self.stats.solves += self.nw * Nrv
        self.stats.iters_sum += int(iters.sum())
        return np.transpose(out, (0,) + tuple(range(out.ndim - 1, 0, -1))), iters

    # ---- EFA-PFF-HMC trajectory on the device (src/EFAPFFHMCUpdater.jl:102-276) ------------------------------
    def efa_setup(self, omega=1.0, mass=1.0):
        """ExactFourierAccelerator(Ω, M, β, Δτ, η = 0) tables (src/EFAPFFHMCUpdater.jl:92): q = m = Δτ M [Ω² + 4/Δτ² sin²(πω/Lτ)] per phonon
        mode; infinite-mass partner modes (bond SSH) are frozen."""
        fm = np.asarray(self.force_couplings.finite_mass, dtype=bool)
        om = np.arange(self.Lt)
        row = self.dtau * mass * (omega**2 + 4.0 / self.dtau**2 * np.sin(np.pi * om / self.Lt) ** 2)
        q = np.asfortranarray(np.where(fm[:, None], row[None, :], np.inf))
        self.efa_q, self.efa_m = q, q.copy(order="F")
        self.h.call("smoqy_efa_config", L.ptr(self.efa_q), L.ptr(self.efa_m))
        self._efa_R = self.h.pinned_empty((self.nw, self.Lt, self.Nph_force))
        self._efa_rv = self.h.pinned_empty((self.Nt, self.nw, self.N))

    def efa_energies(self):
        K, Sb = np.zeros(self.nw), np.zeros(self.nw)
        self.h.call("smoqy_efa_energies", L.ptr(K), L.ptr(Sb))
        return K, Sb

    def hmc_trajectory_device(self, dt=None):
        """hmc_update! (src/EFAPFFHMCUpdater.jl:102-276) with the whole trajectory on the device: Φ sampled (:133), momenta refreshed
        (:142), evolve(Δt/2) + Nt × {force solve, kick, evolve, update!} in ONE library call (:148-206), final action at tol_action (:217),
        ΔH (:234-250).  Returns (ΔH per walker, last action tuple); the caller accepts or rejects (smoqy_efa_checkpoint)."""
        dt = np.pi / (2 * self.Nt) if dt is None else float(dt)          # tutorials/holstein_honeycomb.jl:542
        sf0 = self.sample_pseudofermion_fields()                          # :133
        self.h.call("smoqy_efa_checkpoint", 0)                            # copyto!(x0, x), :130
        d = self._draws
        R = self._efa_R if d is None else d.efa_R
        if d is None:
            list(self.pool.map(lambda w: self.rng[w].standard_normal(out=R[w].reshape(-1)), range(self.nw)))
        K0 = np.zeros(self.nw)
        self.h.call("smoqy_efa_initialize_momentum", L.ptr(R), L.ptr(K0))  # :142
        _, Sb0 = self.efa_energies()                                      # bosonic action, :136
        rv = self._efa_rv if d is None else d.efa_rv
        if d is None:
            for t in range(self.Nt):
                for w in range(self.nw):
                    self.rng[w].standard_normal(out=rv[t, w])             # randn!(rng, v) of each update_preconditioner!, KPMPreconditioner.jl:634
        sf = np.zeros((self.Nt, self.nw))
        iters = np.zeros((self.Nt, self.nw), dtype=np.int32)
        eps = np.zeros((self.Nt, self.nw))
        gate = WalkerBatch.solve_gate
        if gate is not None:
            gate.acquire()
        try:
            self.h.call("smoqy_hmc_trajectory_v", self.phi, self.u, int(self.Nt), C.c_double(dt), C.c_double(self.tol_force), int(self.maxiter), 1, L.ptr(rv), L.ptr(sf), L.ptr(iters), L.ptr(eps))
        finally:
            if gate is not None:
                gate.release()
        self.stats.solves += self.nw * self.Nt
        self.stats.iters_sum += int(iters.sum())
        if self.iter_log is not None:
            self.iter_log.extend(iters[t].copy() for t in range(self.Nt))
        last = self.pff_step(self.tol, moved=False, want_force=False)      # final action, :217
        K1, Sb1 = self.efa_energies()                                      # :238-244
        self.dH = (last[0] + Sb1 + K1) - (sf0 + Sb0 + K0)                   # :247-250
        return self.dH, last

    def drift_fields(self, pis, step):
        np.multiply(pis, step, out=self._tmp)
        np.add(self.xs, self._tmp, out=self.xs)
        self.refresh_fields()

    def drift_by(self, dx):
        """x += dx with a precomputed increment (one pass over the fields)."""
        np.add(self.xs, dx, out=self.xs)
        self.refresh_fields()

    def _momentum(self):
        return np.stack(self._randn_all((self.Lt, self.Nph)))

    # ---- one synthetic sweep -------------------------------------------------------------------------
    def sweep(self):
        if self.prefetch_randoms:
            self._draws = self._take_draws()   # drawn while the previous sweep ran (now, for the first one)
            self._submit_draws()               # the next sweep's, under this sweep's device work
            try:
                return self._sweep()
            finally:
                self._draws = None
        return self._sweep()

    def _sweep(self):
        last = None
        d = self._draws
        # reflection-like and swap-like global moves: sample Φ, move the fields, one action solve
        # (src/reflection_update.jl:69-114, src/swap_update.jl)
        for _ in range(2):
            self.sample_pseudofermion_fields()
            if d is not None:
                dx = d.dx[d.i_dx]              # momentum × drift, formed by the thread that drew it
                d.i_dx += 1
                self.drift_by(dx)
                last = self.calculate_fermionic_action(self.tol)
                np.subtract(self.xs, dx, out=self.xs)  # x + (−drift·π) of the branch below, bit for bit
                self.refresh_fields()
                continue
            pis = self._momentum()
            self.drift_fields(pis, self.drift)
            last = self.calculate_fermionic_action(self.tol)
            self.drift_fields(pis, -self.drift)  # "rejected": restore x, update! (src/reflection_update.jl)
        # HMC trajectory (src/EFAPFFHMCUpdater.jl:102-276)
        if self.device_efa:
            # the real EFA leapfrog, device resident; the move is then ALWAYS rejected (x restored from x0, :263-275) so that the
            # benchmark keeps solving on the field distribution SURVEY.md §8(d) defines instead of thermalising away from it
            _, last = self.hmc_trajectory_device()
            self.h.call("smoqy_efa_checkpoint", 1)
            self.stats.action = float(np.sum(last[0]))
            if self.measure_nrv:
                self.measure_greens(self.measure_nrv)
            return last
        self.sample_pseudofermion_fields()
        pis = self._momentum()
        dx = pis * (self.drift / self.Nt)
        if self.device_update:
            # every force evaluation is one library call; the field move of the previous step rides in front of it
            for t in range(self.Nt):
                self.pff_step(self.tol_force, moved=t > 0, want_force=True)
                np.add(self.xs, dx, out=self.xs)      # stand-in for evolve_eom!
            last = self.pff_step(self.tol, moved=True, want_force=False)
        else:
            for _ in range(self.Nt):
                self.calculate_fermionic_action(self.tol_force)
                self.fermionic_force()
                self.drift_by(dx)
            last = self.calculate_fermionic_action(self.tol)
        self.drift_fields(pis, -self.drift)      # reject: restore x
        self.stats.action = float(np.sum(last[0]))
        if self.measure_nrv:
            self.measure_greens(self.measure_nrv)   # make_measurements! -> update_greens_estimator! once per sweep (tutorials/holstein_honeycomb.jl:653-671)
        return last

    @property
    def solves_per_sweep(self):
        return 2 + self.Nt + 1 + self.measure_nrv


class WalkerTeam:
    """K single-walker control flows on ONE batched handle (C ABI "walker teams", csrc/team.hip).

    The reference runs each walker as its own MPI rank around single-walker objects (tutorials/holstein_honeycomb_mpi.jl:60-72).  A team
    keeps that per-walker code shape — every member below is driven by its own host thread and only ever sees its own walker — while the
    library advances all K walkers with one launch per kernel: a member's call blocks until the other K - 1 members have made the same
    call, the last arrival runs the batched call, everybody returns with its own results.

    ``WalkerTeam(workload, K)`` builds the batched handle exactly as ``WalkerBatch`` does (same lattice tables, couplings, bare model) and
    hands out ``members[w]``; ``member.sweep()`` is the host-driven sweep of ``WalkerBatch.sweep`` (device_efa = False) for ONE walker.
    """

    def __init__(self, workload: str, nwalkers: int, walker0: int = 0, device: int = -1, device_efa: bool = False, **kw):
        # device_efa: configure the exact-Fourier-acceleration state on the handle so that members can call hmc_update (the whole
        # trajectory of a member's hmc_update! on the device)
        self.batch = WalkerBatch(workload, nwalkers=nwalkers, walker0=walker0, device=device, device_efa=device_efa, host_threads=1, **kw)
        b = self.batch
        self._t = C.c_void_p()
        lib = L.load()
        rc = lib.smoqy_team_create(C.byref(self._t), b.h._h, int(b.Nph_force))
        if rc:
            raise L.SmoqyError(f"smoqy_team_create failed ({rc}): " + (lib.smoqy_team_last_error(None) or b"").decode())
        self.lib = lib
        self.members = [TeamMember(self, w) for w in range(nwalkers)]

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self._t, *args)
        if rc:
            raise L.SmoqyError(f"{name} failed ({rc}): " + (self.lib.smoqy_team_last_error(self._t) or b"").decode())

    def ge_config(self, Nrv: int = 10):
        """GreensEstimator for the members (smoqy_team_ge_config): a follower handle with Nrv right-hand sides per walker; call before
        ``serve``.  Lattice geometry as in ``WalkerBatch.measure_greens``."""
        b = self.batch
        meta = b.models[0].meta
        if "L" not in meta:
            raise ValueError("no lattice geometry recorded for this workload")
        Lc = int(meta["L"])
        n_orb = 2 if b.models[0].name.startswith("holstein_honeycomb") else 1
        Ls = (Lc, Lc) if n_orb * Lc * Lc == b.N else (Lc,)
        self.call("smoqy_team_ge_config", int(Nrv), n_orb, len(Ls), L.ptr(np.asarray(Ls, dtype=np.int64)))
        self.ge = {"Nrv": int(Nrv), "Ls": list(Ls)}
        for m in self.members:
            m.ge = self.ge
        return self.ge

    def serve(self, name: str):
        """Publish the team for members of other processes (smoqy_team_serve): ranks join with ``RemoteMember(name, w)``.  The members'
        initial phonon fields are the batch's."""
        b = self.batch
        x0 = np.ascontiguousarray(np.stack([np.asarray(b.xs_force[w]) for w in range(b.nw)]))
        self.call("smoqy_team_serve", name.encode(), L.ptr(x0))
        return {"name": name, "tol": b.tol, "tol_force": b.tol_force, "maxiter": b.maxiter, "Nt": b.Nt, "drift": b.drift, "free": b.Nph, "device_efa": bool(b.device_efa),
                "ge": getattr(self, "ge", None)}

    def unserve(self):
        self.call("smoqy_team_unserve")

    def close(self):
        if self._t:
            self.lib.smoqy_team_destroy(self._t)
            self._t = C.c_void_p()
        self.batch.h.close()


class TeamMember:
    """One walker of a WalkerTeam: the per-walker state a rank of the reference owns (phonon fields x, rng) and the update sequence of
    tutorials/holstein_honeycomb.jl:611-684 written for ONE walker; every library call goes through the team."""

    def __init__(self, team: WalkerTeam, w: int):
        self.team, self.w = team, w
        b = team.batch
        self.Lt, self.N, self.Nph = b.Lt, b.N, b.Nph_force
        self.rng = b.rng[w]
        self.x = np.array(b.xs_force[w], copy=True)          # (Lt, Nph) C order == Nph x Ltau column-major
        self.free = b.Nph                                     # bond-SSH: the last mode is the frozen partner
        self.R = np.empty((self.Lt, self.N), dtype=np.complex128, order="F")
        self.dSdx = np.zeros((self.Lt, self.Nph))
        self.tol, self.tol_force, self.maxiter, self.Nt, self.drift = b.tol, b.tol_force, b.maxiter, b.Nt, b.drift
        self.solves = self.iters_sum = 0

    def _sample_call(self, R, rr):
        self.team.call("smoqy_team_sample_phi", self.w, R, rr)

    def _step_call(self, *a):
        self.team.call("smoqy_team_pff_step", self.w, *a)

    def _hmc_call(self, *a):
        self.team.call("smoqy_team_hmc_update", self.w, *a)

    def _finish_call(self, accept):
        self.team.call("smoqy_team_hmc_finish", self.w, int(accept))

    def _ge_update_call(self, *a):
        self.team.call("smoqy_team_ge_update", self.w, *a)

    def _ge_gd0_call(self, *a):
        self.team.call("smoqy_team_ge_measure_GD0", self.w, *a)

    def measure_greens(self, orbitals=(1, 1), tol=None):
        """update_greens_estimator! + measure_GΔ0! (src/Measurements/GreensEstimator.jl:125-233) for this walker: the member draws its Nrv
        unit-modulus random vectors (:141-142) and the Lanczos start vector; all members' K·Nrv systems advance in one batched CG.  Returns
        (G(Δ,0) of shape (Lτ+1, L...), iterations summed over the Nrv solves)."""
        ge = getattr(self, "ge", None)
        if not ge:
            raise RuntimeError("the team has no GreensEstimator (WalkerTeam.ge_config before serve)")
        Nrv, Ls = ge["Nrv"], tuple(ge["Ls"])
        R = np.empty((self.Lt, self.N, Nrv), dtype=np.complex128, order="F")
        flat = R.reshape(-1, order="F").view(np.float64)
        self.rng.standard_normal(out=flat)
        np.divide(R, np.abs(R), out=R)
        rv = self.rng.standard_normal(self.N)
        it, ep = C.c_int(0), C.c_double(0.0)
        self._ge_update_call(L.ptr(R), L.ptr(rv), C.c_double(self.tol if tol is None else tol), int(self.maxiter), C.byref(it), C.byref(ep))
        out = np.zeros((*reversed(Ls), self.Lt + 1), dtype=np.complex128)   # C order == (Lτ+1) x L... column-major
        self._ge_gd0_call(int(orbitals[0]), int(orbitals[1]), L.ptr(out))
        self.solves += Nrv
        self.iters_sum += it.value
        return np.transpose(out, tuple(range(out.ndim - 1, -1, -1))), it.value

    def hmc_update(self, dt=None, send_x=True, drawn=None):
        """hmc_update! (src/EFAPFFHMCUpdater.jl:102-276) for this walker with the trajectory on the device: the member draws what the
        reference draws from its rng (Φ deviates :133, momenta :142, one Lanczos start vector per solve), the library runs the leapfrog for
        all members at once.  Returns (ΔH, x_new); the caller decides and reports with ``hmc_finish(accept)``.  ``drawn``: (R, P, rvs)
        already drawn from this member's generator (``prefetch_randoms``)."""
        dt = np.pi / (2 * self.Nt) if dt is None else float(dt)          # tutorials/holstein_honeycomb.jl:542
        if drawn is not None:
            R, P, rvs = drawn
        else:
            R = self.R
            self._draw_cn(R)
            P = self.rng.standard_normal((self.Lt, max(self.Nph, 1)))
            rvs = self.rng.standard_normal((self.Nt + 1, self.N))
        H0, H1 = np.zeros(3), np.zeros(3)
        x_new = np.empty_like(self.x)
        it = C.c_int(0)
        self._hmc_call(L.ptr(self.x) if send_x else None, L.ptr(R), L.ptr(P), L.ptr(rvs), int(self.Nt), C.c_double(dt), C.c_double(self.tol_force), C.c_double(self.tol), int(self.maxiter),
                       L.ptr(H0), L.ptr(H1), L.ptr(x_new), C.byref(it))
        self.solves += self.Nt + 1
        self.iters_sum += it.value
        self.H0, self.H1 = H0, H1
        return float(H1.sum() - H0.sum()), x_new

    def hmc_finish(self, accept, x_new=None):
        self._finish_call(accept)
        if accept and x_new is not None:
            self.x[...] = x_new

    def sweep_device_hmc(self):
        """the sweep of ``WalkerBatch.sweep`` with device_efa = True for ONE walker: two local-move-like updates, then hmc_update! with the
        trajectory on the device, always rejected (bench.py's convention: the field distribution stays the one SURVEY.md §8(d) defines)"""
        d = self._take_draws() if getattr(self, "_draw_pool", None) is not None else None
        for i in range(2):
            self.sample_pseudofermion_fields(None if d is None else d["R"][i])
            pi = self.rng.standard_normal((self.Lt, self.free)) if d is None else d["pi"][i]
            self.x[:, : self.free] += self.drift * pi
            self.pff_step(self.tol, moved=True, want_force=False, rv=None if d is None else d["rv"][i])
            self.x[:, : self.free] -= self.drift * pi
        dH, _ = self.hmc_update(drawn=None if d is None else (d["R"][2], d["P"], d["rvs"]))
        self.hmc_finish(False)
        return dH

    # ---- random numbers one sweep ahead (what WalkerBatch(prefetch_randoms=True) does for a lock-step batch) ------------------------
    def prefetch_randoms(self, pool):
        """From now on ``sweep_device_hmc`` takes its random numbers from arrays that ``pool`` (a ThreadPoolExecutor: one per team, or one
        thread of a rank's own) fills one sweep ahead — the member's generator is asked for the same arrays in the same order, while the
        member waits in the team's rendezvous instead of between two calls."""
        self._draw_pool = pool
        self._draw_sets = [{"R": [np.empty((self.Lt, self.N), dtype=np.complex128, order="F") for _ in range(3)], "pi": [np.empty((self.Lt, self.free)) for _ in range(2)],
                            "rv": [np.empty(self.N) for _ in range(2)], "P": np.empty((self.Lt, max(self.Nph, 1))), "rvs": np.empty((self.Nt + 1, self.N))} for _ in range(2)]
        self._draw_n = 0
        self._draw_next = None

    def _fill_draws(self, d):
        for i in range(2):
            self._draw_cn(d["R"][i])
            self.rng.standard_normal(out=d["pi"][i])
            self.rng.standard_normal(out=d["rv"][i])
        self._draw_cn(d["R"][2])
        self.rng.standard_normal(out=d["P"])
        self.rng.standard_normal(out=d["rvs"])
        return d

    def _take_draws(self):
        if self._draw_next is None:
            self._draw_next = self._draw_pool.submit(self._fill_draws, self._draw_sets[self._draw_n & 1])
        d = self._draw_next.result()
        self._draw_n += 1
        self._draw_next = self._draw_pool.submit(self._fill_draws, self._draw_sets[self._draw_n & 1])
        return d

    def _draw_cn(self, R):
        flat = R.reshape(-1, order="F").view(np.float64)
        self.rng.standard_normal(out=flat)                    # randn!(rng, Φ), src/PFFCalculator.jl:67
        flat *= np.sqrt(0.5)

    def sample_pseudofermion_fields(self, R=None):
        if R is None:
            R = self.R
            self._draw_cn(R)
        rr = C.c_double(0.0)
        self._sample_call(L.ptr(R), C.byref(rr))
        return rr.value

    def pff_step(self, tol, moved, want_force, rv=None):
        if rv is None:
            rv = self.rng.standard_normal(self.N)             # randn!(rng, v), KPMPreconditioner.jl:634
        sf, eps, it = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        self._step_call(L.ptr(self.x) if moved else None, L.ptr(rv), C.c_double(tol), int(self.maxiter), 1, C.byref(sf), C.byref(it), C.byref(eps),
                        L.ptr(self.dSdx) if want_force else None)
        self.solves += 1
        self.iters_sum += it.value
        return sf.value, it.value, eps.value

    def sweep(self):
        """reflection-like + swap-like move + HMC(Nt) with the synthetic host-side drift: WalkerBatch.sweep (device_efa = False), one walker"""
        last = None
        for _ in range(2):
            self.sample_pseudofermion_fields()
            pi = self.rng.standard_normal((self.Lt, self.free))
            self.x[:, : self.free] += self.drift * pi
            last = self.pff_step(self.tol, moved=True, want_force=False)
            self.x[:, : self.free] -= self.drift * pi         # "rejected": restore x (the next call sends it)
        self.sample_pseudofermion_fields()
        pi = self.rng.standard_normal((self.Lt, self.free))
        dx = pi * (self.drift / self.Nt)
        for t in range(self.Nt):
            self.pff_step(self.tol_force, moved=True, want_force=True)
            self.x[:, : self.free] += dx
        last = self.pff_step(self.tol, moved=True, want_force=False)
        self.x[:, : self.free] -= self.drift * pi
        return last


class RemoteMember(TeamMember):
    """A member of a team published by ANOTHER process (``WalkerTeam.serve``): what one MPI rank of the reference becomes when the ranks
    of a node share a GPU through one serving rank.  Needs no GPU and no handle — only the library (for the shared-memory rendezvous) and
    the run parameters the serving rank announces (``info``: the dict ``serve`` returned, sent over whatever channel the job has)."""

    def __init__(self, info: dict, w: int, seed: int = 0, wait_seconds: float = 60.0):
        self.lib = L.load_member()  # libsmoqy_member.so: the rendezvous only — no HIP runtime, no rocFFT in this rank
        self._m = C.c_void_p()
        rc = self.lib.smoqy_member_attach(C.byref(self._m), info["name"].encode(), int(w), C.c_double(wait_seconds))
        if rc:
            raise L.SmoqyError(f"smoqy_member_attach failed ({rc}): " + (self.lib.smoqy_member_last_error(None) or b"").decode())
        d = (C.c_int * 4)()
        self.lib.smoqy_member_dims(self._m, d)
        self.team, self.w = None, w
        self.Lt, self.N, self.K, self.Nph = d[0], d[1], d[2], d[3]
        self.rng = np.random.default_rng([seed, w])
        self.x = np.zeros((self.Lt, max(self.Nph, 1)))
        self.lib.smoqy_member_fields(self._m, L.ptr(self.x))
        self.free = int(info["free"])
        self.R = np.empty((self.Lt, self.N), dtype=np.complex128, order="F")
        self.dSdx = np.zeros((self.Lt, self.Nph))
        self.tol, self.tol_force, self.maxiter, self.Nt, self.drift = info["tol"], info["tol_force"], info["maxiter"], info["Nt"], info["drift"]
        self.solves = self.iters_sum = 0
        self.ge = info.get("ge")

    def _call(self, name, *a):
        rc = getattr(self.lib, name)(self._m, *a)
        if rc:
            raise L.SmoqyError(f"{name} failed ({rc}): " + (self.lib.smoqy_member_last_error(self._m) or b"").decode())

    def _sample_call(self, R, rr):
        self._call("smoqy_member_sample_phi", R, rr)

    def _step_call(self, *a):
        self._call("smoqy_member_pff_step", *a)

    def _hmc_call(self, *a):
        self._call("smoqy_member_hmc_update", *a)

    def _finish_call(self, accept):
        self._call("smoqy_member_hmc_finish", int(accept))

    def _ge_update_call(self, *a):
        self._call("smoqy_member_ge_update", *a)

    def _ge_gd0_call(self, *a):
        self._call("smoqy_member_ge_measure_GD0", *a)

    def close(self):
        if self._m:
            self.lib.smoqy_member_detach(self._m)
            self._m = C.c_void_p()
