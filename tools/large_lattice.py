"""Functional check at lattices larger than the headline one, up to and beyond the LDS limits: python tools/large_lattice.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd.walkers import WalkerBatch
lat = sq.lattice
cases = [("honeycomb L=32 Lt=64", lambda **kw: lat.holstein_honeycomb(32, 64, **kw)), ("honeycomb L=36 Lt=32", lambda **kw: lat.holstein_honeycomb(36, 32, **kw)),
         ("honeycomb L=48 Lt=32", lambda **kw: lat.holstein_honeycomb(48, 32, **kw)), ("square L=64 Lt=32", lambda **kw: lat.ossh_square(64, 32, **kw)),
         ("chain L=8192 Lt=32", lambda **kw: lat.bssh_chain(8192, 32, **kw)), ("honeycomb L=72 Lt=16", lambda **kw: lat.holstein_honeycomb(72, 16, **kw))]
for name, fn in cases:
    lat.CONFIGS["_tmp"] = fn
    for pre in (True, False):
        try:
            b = WalkerBatch("_tmp", nwalkers=1)
            sf0 = b.sample_pseudofermion_fields()
            t0 = time.perf_counter(); sf, it, eps = b.calculate_fermionic_action(1e-8, use_precond=pre); dt = time.perf_counter() - t0
            f = b.fermionic_force()
            print(f"{name} precond={pre}: N={b.N} iters {it} eps {eps.max():.1e} |S-|R|^2|/S {np.abs(sf - sf0).max() / sf0.max():.1e} solve {1e3*dt:.1f} ms force ok {np.isfinite(f).all()}", flush=True)
            b.h.close()
        except Exception as e:
            print(f"{name} precond={pre}: {type(e).__name__}: {str(e)[:160]}", flush=True)
