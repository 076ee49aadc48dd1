#!/usr/bin/env python3
"""The call sequence of the reference tutorial (tutorials/holstein_honeycomb.jl:472-684), written against the Python mirror of
the operator API — what a user of SmoQyElPhQMC.jl would recognise after switching the backend:

    SymFermionDetMatrix -> KPMPreconditioner -> PFFCalculator -> sample_pseudofermion_fields! ->
    calculate_derivative_fermionic_action! along a short trajectory -> GreensEstimator / measure_GΔ0!

The phonon fields are a synthetic Holstein configuration and the "leapfrog" is a plain drift (SmoQyDQMC's EFA integrator is
not part of this repository); every solve, force and contraction runs on the MI355X.

    python examples/holstein_honeycomb_demo.py [L] [Ltau]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import smoqyelphqmc_amd as sq  # noqa: E402

L, Lt = (int(sys.argv[1]) if len(sys.argv) > 1 else 6), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
rng = np.random.default_rng(1)
model = sq.lattice.holstein_honeycomb(L, Lt)                       # model geometry, couplings, x ~ N(0,1)
fpi, elph = model.fpi, model.elph

fdm = sq.SymFermionDetMatrix(fpi, maxiter=10_000, tol=1e-10)       # tutorials/holstein_honeycomb.jl:472-480
P = sq.KPMPreconditioner(fdm, rng=rng, rbuf=0.10, n=20, a1=1.0, a2=1.0)
pff = sq.PFFCalculator(elph, fdm)
sq.set_force_couplings(fdm, model.force_couplings(fdm.checkerboard_perm))
print(f"honeycomb L = {L}: N = {fdm.N} sites, Ltau = {fdm.Lt}; preconditioner active = {P.active}, bounds = {P.bounds}, max order = {max(P.order)}")

Sf0 = sq.sample_pseudofermion_fields(pff, elph, fdm, rng=rng)      # Φ = Λᵀ Mᵀ R, returns |R|²
print(f"sampled Φ: S_f = |R|² = {Sf0:.6f}")

dt, pi = 0.02, rng.standard_normal(elph.x.shape)
for step in range(4):                                              # a few HMC-like steps
    dSdx = np.zeros(elph.x.shape, order="F")
    Sf, iters, eps = sq.calculate_derivative_fermionic_action(dSdx, pff, elph, fdm, P, rng, tol=1e-5, maxiter=10_000)
    print(f"step {step}: S_f = {Sf:.6f}  ({iters} CG iterations, residual {eps:.1e}), |dS_f/dx|_max = {np.abs(dSdx).max():.4f}")
    elph.x[...] += dt * pi                                         # stand-in for evolve_eom!
    model.refresh_from_x()                                         # SmoQyDQMC.update!(fermion_path_integral, elph, x, ±1)
    sq.update(fdm, fpi)                                            # update!(fermion_det_matrix, fermion_path_integral)
    fdm._force_couplings.x[...] = elph.x

ge = sq.GreensEstimator(fdm, (2, (L, L)), Nrv=10, preconditioner=P, rng=rng, maxiter=10_000, tol=1e-10)
G = np.zeros((L, L, fdm.Lt + 1), dtype=complex)
sq.measure_GΔ0(G, ge, (1, 1))
print(f"GreensEstimator (Nrv = 10): G_AA(r=0, τ=0) = {G[0, 0, 0].real:.4f}, G_AA(r=0, τ=β) = {G[0, 0, -1].real:.4f}  (sum = {G[0, 0, 0].real + G[0, 0, -1].real:.4f}, exactly 1)")
