"""Where the HOST time of a one-walker sweep goes (cProfile over 6 sweeps, prefetch on): python tools/history/host_profile.py [walkers]"""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
from smoqyelphqmc_amd.walkers import WalkerBatch
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nw, device_efa=True, prefetch_randoms=True)
b.sweep(); b.sweep()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    b.sweep()
b.h.call("smoqy_sync")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print(s.getvalue())
