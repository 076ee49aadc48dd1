"""Launch only the fused MᵀM kernel (for rocprofv3 --pmc passes): `python tools/matvec_only.py [batch] [reps] [workload]`."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wl = sys.argv[3] if len(sys.argv) > 3 else "holstein_honeycomb_L16_Ltau128"
b = WalkerBatch(wl, nwalkers=nb)   # the handle bench.py builds: fields formed on the device from the phonon fields
h = b.h
if os.environ.get("SMOQY_TC"):
    h.call("smoqy_set_tau_chunk", int(os.environ["SMOQY_TC"]))  # τ-chunk override for A/B runs
va, vb = h.vec_alloc(), h.vec_alloc()
g = np.random.default_rng(0)
h.vec_upload(va, np.asfortranarray(g.standard_normal((b.Lt, b.N, nb)) + 1j * g.standard_normal((b.Lt, b.N, nb))))
ms_t = h.bench_matvec(L.OP_MTM, vb, va, reps)
print(f"batch {nb}: {ms_t / reps * 1e3:.2f} us per MtM launch, {h.algorithmic_bytes(L.OP_MTM) / (ms_t / reps * 1e-3) / 1e9:.0f} GB/s algorithmic, {h.describe()['mtm']}, algorithmic_bytes {h.algorithmic_bytes(L.OP_MTM):.0f}")
