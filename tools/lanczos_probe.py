"""Per-step cost of the Lanczos kernel: update_preconditioner! with n Lanczos steps (smoqy_precond_config), many times, for a rocprofv3 --stats pass.
usage: python tools/lanczos_probe.py <n_lanczos> [walkers] [workload]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
n = int(sys.argv[1]); nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = sys.argv[3] if len(sys.argv) > 3 else "holstein_honeycomb_L16_Ltau128"
b = WalkerBatch(wl, nwalkers=nw)
b.h.call("smoqy_precond_config", C.c_double(0.10), n, C.c_double(2.0), C.c_double(1.0))
g = np.random.default_rng(1)
rv = np.ascontiguousarray(g.standard_normal((nw, b.N)))
for _ in range(100):
    b.h.call("smoqy_precond_update_all", L.ptr(rv))
b.h.call("smoqy_sync")
print("done", n)
