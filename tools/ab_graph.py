import sys, time
sys.path.insert(0, '.')
import torch
from smoqyelphqmc_amd.walkers import WalkerBatch
for nw in (1, 4):
    for g in (0, 1, 0, 1):
        b = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nw)
        b.h.call("smoqy_cg_use_graph", g)
        b.sweep(); b.sweep()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); b.sweep(); b.h.call("smoqy_sync"); ts.append(time.perf_counter() - t0)
        print(f"walkers {nw:2d} graph={'on ' if g else 'off'}: sweep ms {[round(1e3*t,1) for t in ts]}", flush=True)
        b.h.close()
