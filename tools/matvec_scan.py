"""Fused MᵀM timing vs batch and tau-chunk: `python tools/matvec_scan.py [batches...]` (env SMOQY_FDM_OWN=0 for the LDS-resident kernel)."""
import sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L

lat = sq.lattice
batches = [int(x) for x in sys.argv[1:]] or [1, 4, 16, 64]
for nb in batches:
    ms = [lat.holstein_honeycomb(16, 128, walker=w) for w in range(min(nb, 4))]
    nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
    h = L.Handle(128, 512, nt, colors, True, nb, 1, -1)
    for w in range(nb):
        m = ms[w % len(ms)]
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    a, b = h.vec_alloc(), h.vec_alloc()
    g = np.random.default_rng(0)
    h.vec_upload(a, np.asfortranarray(g.standard_normal((128, 512, nb)) + 1j * g.standard_normal((128, 512, nb))))
    out = []
    for tc in (1, 2):
        h.call("smoqy_set_tau_chunk", tc)
        ms_t = h.bench_matvec(L.OP_MTM, b, a, 200)
        out.append(f"Tc={tc}: {ms_t / 200 * 1e3:6.2f} us {h.algorithmic_bytes(L.OP_MTM) / (ms_t / 200 * 1e-3) / 1e9:6.0f} GB/s")
    print(f"batch {nb:3d}  " + "   ".join(out), flush=True)
    del h
