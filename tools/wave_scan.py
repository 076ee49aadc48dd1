"""fused MᵀM: the workgroup kernels (smoqy_matvec_wave 0) against fdm_wave_kernel at several run lengths and batch sizes.
usage: python tools/wave_scan.py [workload] [batches] [run lengths]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch

wl = sys.argv[1] if len(sys.argv) > 1 else "holstein_honeycomb_L16_Ltau128"
batches = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,16,128").split(",")]
runs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,-1,2,4,8,16,32").split(",")]
Tc = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # τ-chunk of the handle (0 = the library's choice); runs are multiples of it
for nw in batches:
    b = WalkerBatch(wl, nwalkers=nw)
    h = b.h
    if Tc:
        h.call("smoqy_set_tau_chunk", Tc)
    g = np.random.default_rng(3)
    va, vb = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(va, np.asfortranarray(g.standard_normal((b.Lt, b.N, nw)) + 1j * g.standard_normal((b.Lt, b.N, nw))))
    row = []
    for R in runs:
        h.call("smoqy_matvec_wave", R)
        h.bench_matvec(L.OP_MTM, vb, va, 20)
        us = h.bench_matvec(L.OP_MTM, vb, va, 200) / 200 * 1e3
        row.append(f"R={R}: {us:.2f} us [{h.describe()['mtm'][:18]}]")
    print(f"{wl} nsys={nw}: " + " | ".join(row), flush=True)
    h.close()
