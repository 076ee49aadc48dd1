import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch
from smoqyelphqmc_amd.walkers import WalkerBatch
b = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=8)
b.sweep()
pr = cProfile.Profile(); pr.enable()
t0=time.perf_counter(); b.sweep(); b.h.call("smoqy_sync"); t1=time.perf_counter()
pr.disable()
print("sweep s:", t1-t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
