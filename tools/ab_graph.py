import sys, time, os
sys.path.insert(0, '.')
import torch
from smoqyelphqmc_amd.walkers import WalkerBatch
for nw in (1, 16):
    for ng in ("0", "1"):
        os.environ["SMOQY_NO_GRAPH"] = ng
        b = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nw)
        b.sweep()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); b.sweep(); b.h.call("smoqy_sync"); ts.append(time.perf_counter() - t0)
        print(f"walkers {nw:2d} graph={'off' if ng=='1' else 'on '}: sweep ms {[round(1e3*t,1) for t in ts]}", flush=True)
        b.h.close()
