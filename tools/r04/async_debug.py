import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
# dirty the device heap: what a long test session leaves behind
junk = [torch.full((1 << 28,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(8)]   # 16 GiB of NaN
del junk
torch.cuda.empty_cache()
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from test_gpu_efa import _drive, _async_counts
name = sys.argv[1] if len(sys.argv) > 1 else "ossh_square_L12_Ltau100_alpha0p2"
nw, Nt, dt, tol = 2, 6, 0.09, 1e-6
a = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
bb = WalkerBatch(name, nwalkers=nw, device_efa=True, Nt=Nt)
bb.h.call("smoqy_hmc_async", 0, None, None)
def pstate(b, w):
    act = C.c_int(0); bounds = np.zeros(2); order = np.zeros(b.Lt, dtype=np.int32); n = C.c_int(0); la, lb = np.zeros(20), np.zeros(19)
    b.h.call("smoqy_precond_get", w, C.byref(act), bounds.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(n), la.ctypes.data_as(C.POINTER(C.c_double)), lb.ctypes.data_as(C.POINTER(C.c_double)))
    return act.value, bounds.round(6).tolist(), order[:6].tolist(), int(order[:n.value].sum())
g = np.random.default_rng(17)
for trip in range(3):
    Rphi = np.asfortranarray((g.standard_normal((a.Lt, a.N, nw)) + 1j * g.standard_normal((a.Lt, a.N, nw))) * np.sqrt(0.5))
    R = np.ascontiguousarray(g.standard_normal((nw, a.Lt, a.Nph_force)))
    rv = np.ascontiguousarray(g.standard_normal((Nt, nw, a.N)))
    oa = _drive(a, Rphi, R, rv, Nt, dt, tol); ob = _drive(bb, Rphi, R, rv, Nt, dt, tol)
    print("trip", trip, "async counts", _async_counts(a))
    print(" iters A", oa[1].T.tolist()); print(" iters B", ob[1].T.tolist())
    print(" sf diff", np.abs(oa[0] - ob[0]).max(), "x diff", np.abs(oa[3] - ob[3]).max())
    for w in range(nw):
        print("  pre A", w, pstate(a, w)); print("  pre B", w, pstate(bb, w))
    for b in (a, bb):
        b.h.call("smoqy_efa_checkpoint", 1)
