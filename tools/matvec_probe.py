"""Probe: MtM / M kernel time vs batch and tau-chunk.  `python tools/matvec_probe.py`"""
import sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
lat = sq.lattice
for nb in (1, 16, 64):
    ms = [lat.holstein_honeycomb(16, 128, walker=w) for w in range(min(nb, 4))]
    nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
    h = L.Handle(128, 512, nt, colors, True, nb, 1, -1)
    for w in range(nb):
        m = ms[w % len(ms)]
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(m.fpi.t), L.ptr(perm), C.c_double(m.fpi.dtau))
    a, b = h.vec_alloc(), h.vec_alloc()
    g = np.random.default_rng(0)
    h.vec_upload(a, np.asfortranarray(g.standard_normal((128, 512, nb)) + 1j * g.standard_normal((128, 512, nb))))
    for tc in (1, 2):
        h.call("smoqy_set_tau_chunk", tc)
        for op, name in ((L.OP_M, "M"), (L.OP_MTM, "MtM")):
            h.bench_matvec(op, b, a, 20)
            t = h.bench_matvec(op, b, a, 200) / 200 * 1e3
            print(f"batch {nb:3d} Tc {tc} {name:4s}: {t:7.2f} us", flush=True)
    h.close()
