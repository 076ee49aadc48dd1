"""T = ComplexF64 against real T on the benchmarked lattice: fused MtM launch, preconditioner apply, CG solve (time per iteration).
usage: python tools/complex_scan.py [walkers] [L] [Ltau]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
lat = sq.lattice
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Ls = int(sys.argv[2]) if len(sys.argv) > 2 else 16
Lt = int(sys.argv[3]) if len(sys.argv) > 3 else 128
for cplx in (False, True):
    ms = [lat.holstein_honeycomb(Ls, Lt, walker=w) for w in range(nw)]
    nt, perm, colors = lat.checkerboard_decomposition(ms[0].fpi.neighbor_table)
    N = ms[0].fpi.N
    h = L.Handle(Lt, N, nt, colors, True, nw, 1, is_complex=cplx)
    for w, m in enumerate(ms):
        t = m.fpi.t
        if cplx:
            g = np.random.default_rng(300 + w)
            t = np.asfortranarray(t * np.exp(1j * g.uniform(0, 0.3, t.shape[0]))[:, None])
        h.call("smoqy_update_from_path_integral", w, L.ptr(m.fpi.V), L.ptr(t), L.ptr(perm), C.c_double(m.fpi.dtau))
    g = np.random.default_rng(3)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    a, b = h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    h.bench_matvec(L.OP_MTM, b, a, 20)
    us_mtm = h.bench_matvec(L.OP_MTM, b, a, 200) / 200 * 1e3
    rv = np.ascontiguousarray(g.standard_normal((nw, N * (2 if cplx else 1))))
    h.call("smoqy_precond_update_all", L.ptr(rv))
    h.call("smoqy_precond_apply_v", b, a); h.call("smoqy_sync")
    t0 = time.perf_counter()
    for _ in range(100):
        h.call("smoqy_precond_apply_v", b, a)
    h.call("smoqy_sync")
    us_pre = (time.perf_counter() - t0) / 100 * 1e6
    it, eps = np.zeros(nw, dtype=np.int32), np.zeros(nw)
    h.vec_upload(b, v)
    h.call("smoqy_cg_solve_v", b, b, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
    h.vec_upload(b, v); h.call("smoqy_sync")
    t0 = time.perf_counter()
    h.call("smoqy_cg_solve_v", b, b, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
    dt = time.perf_counter() - t0
    print(f"complex T = {cplx}: honeycomb L{Ls} Ltau{Lt} x {nw} walkers: MtM {us_mtm:.1f} us [{h.describe()}], precond apply {us_pre:.1f} us, CG {it.max()} iterations, {dt / it.max() * 1e6:.1f} us per iteration", flush=True)
    h.close()
