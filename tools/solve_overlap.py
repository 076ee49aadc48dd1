"""Throughput of the preconditioned CG solve alone (no field moves, no force, no RNG) with 1, 2, 4 streams of 16 walkers."""
import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
from smoqyelphqmc_amd.walkers import WalkerBatch

per = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bs = [WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=per, walker0=per * s) for s in range(4)]
for b in bs:
    b.sample_pseudofermion_fields()
    b.calculate_fermionic_action(1e-5)
    b.calculate_fermionic_action(1e-5)
NS = 12

def work(b, pre=True):
    for _ in range(NS):
        b.calculate_fermionic_action(1e-5, use_precond=pre) if not pre else _solve_only(b)
    b.h.call("smoqy_sync")

def _solve_only(b):
    # the solve without update_preconditioner! (fields do not move here): Λ⁻ᵀΦ, CG, Λ⁻¹, dot
    import ctypes as C
    from smoqyelphqmc_amd import _lib as L
    b.h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIVT, b.u, b.phi)
    it = np.zeros(b.nw, dtype=np.int32); ep = np.zeros(b.nw)
    b.h.call("smoqy_cg_solve_v", b.u, b.u, C.c_double(1e-5), 10000, 1, L.ptr(it), L.ptr(ep))
    b.stats.iters_sum += int(it.sum()); b.stats.solves += b.nw

for S in (1, 2, 4):
    for b in bs: b.stats.iters_sum = b.stats.solves = 0
    t = time.perf_counter()
    th = [threading.Thread(target=work, args=(bs[s],)) for s in range(S)]
    [x.start() for x in th]; [x.join() for x in th]
    dt = time.perf_counter() - t
    its = sum(b.stats.iters_sum for b in bs[:S]) / max(1, sum(b.stats.solves for b in bs[:S]))
    print(f"{S} stream(s) x {per} walkers: {S * NS / dt:7.1f} batch-solves/s, {1e6 * dt / (NS * its):7.1f} us per CG iteration of one stream, {1e6 * dt / (S * NS * its):7.1f} us per batch-iteration overall ({its:.1f} iters)", flush=True)
