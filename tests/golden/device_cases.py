"""Fixed CG cases whose EXACT device iteration counts are pinned in tests/golden/device_cg_iterations.json (the
device path is deterministic — fixed-order reductions, no atomics — so a tolerance of ±2 against the oracle would hide drift).

`compute()` runs every case through the C ABI and returns {case name: [iterations per system]}.  It is used by
tests/test_golden.py::test_device_iteration_counts_are_pinned (compare) and by `python tests/golden/device_cases.py` on a GPU box
(regenerate: prints the JSON to stdout).  The oracle's counts for the same inputs are stored next to the device's for reference; the two
may differ by one step where the stop test lands within rounding of the tolerance (different summation order)."""
import ctypes as C
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
JSON_PATH = os.path.join(HERE, "device_cg_iterations.json")

BENCH_SHAPES = [("holstein_honeycomb_L16_Ltau128", 16), ("holstein_honeycomb_L16_Ltau128", 8), ("ossh_square_L12_Ltau100", 16), ("bssh_chain_L256_Ltau200", 16),
                ("ossh_square_L12_Ltau100_alpha0p2", 16), ("bssh_chain_L256_Ltau200_alpha0p2", 16)]


def compute(with_oracle=False):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from smoqyelphqmc_amd import _lib as L
    from smoqyelphqmc_amd.walkers import WalkerBatch

    out, ref = {}, {}
    # the three committed fixtures, plain and KPM-preconditioned, Sym and Asym
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        d = dict(np.load(path, allow_pickle=False))
        name = os.path.basename(path)[:-4]
        Lt, N = d["v"].shape
        for is_sym in (True, False):
            tag = "sym" if is_sym else "asym"
            h = L.Handle(Lt, N, d["sorted_table"], d["colors"], is_sym, 1, 1)
            h.call("smoqy_update_from_path_integral", 0, L.ptr(np.asfortranarray(d["V"])), L.ptr(np.asfortranarray(d["t"])), L.ptr(d["perm"]), C.c_double(float(d["dtau"])))
            b = np.asfortranarray(d["b"][:, :, None])
            x = np.zeros_like(b)
            it, eps = np.zeros(1, dtype=np.int32), np.zeros(1)
            h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, 1, C.c_double(1e-10), 20000, 0, L.ptr(it), L.ptr(eps))
            out[f"{name}/{tag}/plain/1e-10"] = it.tolist()
            h.call("smoqy_precond_update", 0, L.ptr(np.ascontiguousarray(d["randvec"])))
            h.call("smoqy_cg_solve", L.ptr(x), L.ptr(b), 1, 0, 1, C.c_double(1e-10), 20000, 1, L.ptr(it), L.ptr(eps))
            out[f"{name}/{tag}/kpm/1e-10"] = it.tolist()
            ref[f"{name}/{tag}/plain/1e-10"] = [int(d[f"oracle_{tag}_cg_iters_plain"])]
            ref[f"{name}/{tag}/kpm/1e-10"] = [int(d[f"oracle_{tag}_cg_iters_kpm"])]
            h.close()
    # the launch shapes bench.py times (same inputs as tests/test_gpu_bench_shape.py::test_pcg_at_the_benchmarked_shape)
    for wl, nw in BENCH_SHAPES:
        batch = WalkerBatch(wl, nwalkers=nw)
        h = batch.h
        rv = np.ascontiguousarray(np.random.default_rng(41).standard_normal((nw, batch.N)))
        h.call("smoqy_precond_update_all", L.ptr(rv))
        g = np.random.default_rng(42)
        shape = (batch.Lt, batch.N, nw)
        bv = np.asfortranarray(g.standard_normal(shape) + 1j * g.standard_normal(shape))
        xb = h.vec_alloc()
        for tol in (1e-10, 1e-5):
            h.vec_upload(xb, bv)
            it, eps = np.zeros(nw, dtype=np.int32), np.zeros(nw)
            h.call("smoqy_cg_solve_v", xb, xb, C.c_double(tol), 10000, 1, L.ptr(it), L.ptr(eps))
            out[f"{wl}/{nw}sys/kpm/{tol:g}"] = it.tolist()
            if with_oracle:
                from oracle import oracle as orc

                its = []
                for w in (0, nw - 1):
                    m = batch.models[w]
                    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, batch.perm, m.fpi.dtau, True)
                    o = orc.OracleFDM(batch.nt, expV, ch, sh, True)
                    P = orc.OracleKPM(o)
                    P.update(rv[w])
                    its.append(o.cg_solve(bv[:, :, w], precond=P, tol=tol, maxiter=10000)[1])
                ref[f"{wl}/{nw}sys/kpm/{tol:g}"] = {"systems": [0, nw - 1], "iterations": its}
        h.close()
    return out, ref


if __name__ == "__main__":
    dev, ref = compute(with_oracle=True)
    print(json.dumps({"_comment": "exact CG iteration counts of the HIP path on MI355X (regenerate: python tests/golden/device_cases.py on a GPU box); 'oracle' = the CPU oracle's counts for the same inputs",
                      "device": dev, "oracle": ref}, indent=1))
