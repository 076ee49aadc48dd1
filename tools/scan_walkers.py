"""Exploration aid: sweeps/s versus walkers per GPU and concurrent streams (one WalkerBatch per
stream, driven by host threads).  Not part of the product or of bench.py's contract."""
import sys, time
sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
import torch  # noqa: F401  (same HIP runtime as bench.py)
from smoqyelphqmc_amd.walkers import WalkerBatch

workload = sys.argv[1] if len(sys.argv) > 1 else "holstein_honeycomb_L16_Ltau128"
combos = [(1, 1), (16, 1), (64, 4)]
for wpg, S in combos:
    bs = [WalkerBatch(workload, nwalkers=wpg // S, walker0=s * (wpg // S)) for s in range(S)]
    def run(n):
        if S == 1:
            for _ in range(n): bs[0].sweep()
        else:
            with ThreadPoolExecutor(S) as ex:
                list(ex.map(lambda b: [b.sweep() for _ in range(n)], bs))
        for b in bs: b.h.call("smoqy_sync")
    run(2)
    t0 = time.perf_counter(); n = 2; run(n); dt = time.perf_counter() - t0
    it = sum(b.stats.iters_sum for b in bs) / max(1, sum(b.stats.solves for b in bs))
    print(f"walkers/gpu {wpg:3d} streams {S}: {wpg * n / dt:8.2f} sweeps/s   {dt / n * 1e3:8.1f} ms per lock-step sweep   avg iters {it:.1f}", flush=True)
    for b in bs: b.h.close()
