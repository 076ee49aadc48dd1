"""Functional check at lattices larger than the headline one: python tools/large_lattice.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd.walkers import WalkerBatch
lat = sq.lattice
for name, fn in (("honeycomb L=24 Lt=160", lambda **kw: lat.holstein_honeycomb(24, 160, **kw)), ("honeycomb L=32 Lt=128", lambda **kw: lat.holstein_honeycomb(32, 128, **kw)),
                 ("square L=32 Lt=100", lambda **kw: lat.ossh_square(32, 100, **kw))):
    lat.CONFIGS["_tmp"] = fn
    try:
        b = WalkerBatch("_tmp", nwalkers=2)
        sf0 = b.sample_pseudofermion_fields()
        t0 = time.perf_counter(); sf, it, eps = b.calculate_fermionic_action(1e-10); dt = time.perf_counter() - t0
        print(f"{name}: N={b.N} iters {it} eps {eps.max():.1e} |S-|R|^2|/S {np.abs(sf - sf0).max() / sf0.max():.1e} solve {1e3*dt:.1f} ms", flush=True)
        b.h.close()
    except Exception as e:
        print(f"{name}: {type(e).__name__}: {e}", flush=True)
