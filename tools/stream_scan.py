"""fused MᵀM: chunked kernel vs the streaming kernel at several run lengths and batch sizes (usage: stream_scan.py [batches] [runs])"""
import sys
sys.path.insert(0, '.')
import numpy as np
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,64,128").split(",")]
runs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,2,4,8,16,32,64").split(",")]
wl = sys.argv[3] if len(sys.argv) > 3 else "holstein_honeycomb_L16_Ltau128"
for nb in batches:
    b = WalkerBatch(wl, nwalkers=nb)
    h = b.h
    g = np.random.default_rng(1)
    va, vb = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(va, np.asfortranarray(g.standard_normal((b.Lt, b.N, nb)) + 1j * g.standard_normal((b.Lt, b.N, nb))))
    row = []
    for R in runs:
        h.call("smoqy_matvec_stream", R)
        h.bench_matvec(L.OP_MTM, vb, va, 30)
        us = h.bench_matvec(L.OP_MTM, vb, va, 300) / 300 * 1e3
        row.append(f"R={R}: {us:.1f}")
    print(f"batch {nb}: " + "  ".join(row) + " us", flush=True)
    h.close()
