"""Do kernels of two HIP streams overlap?  Stream A: preconditioner applies (tau-FFT, latency-bound Chebyshev chain, tau-FFT);
stream B: fused MtM launches.  Wall time of both together against each alone."""
import sys, time, threading
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
A = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nb)
B = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nb, walker0=nb)
A.update_preconditioner(); B.update_preconditioner()
va, vb, vc = A.h.vec_alloc(), B.h.vec_alloc(), B.h.vec_alloc()
g = np.random.default_rng(0)
x = np.asfortranarray(g.standard_normal((128, 512, nb)) + 1j * g.standard_normal((128, 512, nb)))
A.h.vec_upload(va, x); B.h.vec_upload(vb, x)
NA, NB = 400, 1500

def runA():
    for _ in range(NA):
        A.h.call("smoqy_precond_apply_v", va, va)
    A.h.call("smoqy_sync")

def runB():
    B.h.bench_matvec(L.OP_MTM, vc, vb, NB)

for f in (runA, runB): f()
t = time.perf_counter(); runA(); ta = time.perf_counter() - t
t = time.perf_counter(); runB(); tb = time.perf_counter() - t
t = time.perf_counter()
th = [threading.Thread(target=runA), threading.Thread(target=runB)]
[x.start() for x in th]; [x.join() for x in th]
tab = time.perf_counter() - t
print(f"batch {nb}: A alone {1e3*ta:.1f} ms ({1e6*ta/NA:.1f} us per apply), B alone {1e3*tb:.1f} ms ({1e6*tb/NB:.1f} us per MtM), together {1e3*tab:.1f} ms  (sum {1e3*(ta+tb):.1f}, max {1e3*max(ta,tb):.1f})")
