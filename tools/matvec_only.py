"""Launch only the fused MᵀM kernel (for rocprofv3 --pmc passes): `python tools/matvec_only.py [batch] [reps]`."""
import sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lat = sq.lattice
ms = [lat.holstein_honeycomb(16, 128, walker=w) for w in range(nb)]
nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
h = L.Handle(128, 512, nt, colors, True, nb, 1, -1)
for w, m in enumerate(ms):
    h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
import os
if os.environ.get("SMOQY_TC"):
    h.call("smoqy_set_tau_chunk", int(os.environ["SMOQY_TC"]))  # τ-chunk override for A/B runs
a, b = h.vec_alloc(), h.vec_alloc()
g = np.random.default_rng(0)
h.vec_upload(a, np.asfortranarray(g.standard_normal((128, 512, nb)) + 1j * g.standard_normal((128, 512, nb))))
ms_t = h.bench_matvec(L.OP_MTM, b, a, reps)
print(f"batch {nb}: {ms_t / reps * 1e3:.2f} us per MtM launch, {h.algorithmic_bytes(L.OP_MTM) / (ms_t / reps * 1e-3) / 1e9:.0f} GB/s algorithmic")
