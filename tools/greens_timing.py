"""Timing of the batched update_greens_estimator! + measure_GΔ0! at the headline lattice (Nrv = 10)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import smoqyelphqmc_amd as sq
lat = sq.lattice
m = lat.holstein_honeycomb(16, 128)
fdm = sq.SymFermionDetMatrix(m.fpi, maxiter=10000, tol=1e-10)
P = sq.KPMPreconditioner(fdm, rng=np.random.default_rng(0))
rng = np.random.default_rng(1)
ge = sq.GreensEstimator(fdm, (2, (16, 16)), Nrv=10, preconditioner=P, rng=rng, tol=1e-10, maxiter=10000)
for _ in range(2):
    t0 = time.perf_counter(); it = sq.update_greens_estimator(ge, fdm, preconditioner=P, rng=rng, tol=1e-10, maxiter=10000); t1 = time.perf_counter()
    corr = np.zeros((16, 16, 129), dtype=complex)
    for a in (1, 2):
        for b in (1, 2):
            sq.measure_GΔ0(corr, ge, (a, b))
    t2 = time.perf_counter()
    print(f"update_greens_estimator (Nrv=10, avg iters {it:.1f}): {1e3*(t1-t0):.1f} ms;  4 x measure_GD0: {1e3*(t2-t1):.2f} ms", flush=True)
