"""one-stream sweeps/s versus the number of CG pipeline parts: python tools/split_scan.py"""
import sys, time
sys.path.insert(0, '.')
from smoqyelphqmc_amd.walkers import WalkerBatch
for nw in (4, 8, 16, 32):
    row = []
    for parts in (1, 2, 3, 4):
        b = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nw, device_efa=True, cg_split=parts)
        b.sweep(); b.h.call("smoqy_sync")
        t0 = time.perf_counter()
        for _ in range(2):
            b.sweep()
        b.h.call("smoqy_sync")
        row.append(nw * 2 / (time.perf_counter() - t0))
        b.h.close()
    print(nw, "walkers, parts 1..4:", " ".join(f"{x:7.1f}" for x in row), flush=True)
