import sys, time
sys.path.insert(0, '.')
import torch
from smoqyelphqmc_amd.walkers import WalkerBatch
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "holstein_honeycomb_L16_Ltau128"
b = WalkerBatch(wl, nwalkers=nw)
b.sweep()
import smoqyelphqmc_amd._lib as L
import collections
acc = collections.defaultdict(lambda: [0, 0.0])
orig = L.Handle.call
def timed(self, name, *a):
    t = time.perf_counter(); r = orig(self, name, *a); d = time.perf_counter() - t
    acc[name][0] += 1; acc[name][1] += d
    return r
L.Handle.call = timed
b.stats.solves = b.stats.iters_sum = 0
t0 = time.perf_counter(); b.sweep(); b.h.call("smoqy_sync"); t1 = time.perf_counter()
print(f"walkers {nw}: sweep {1e3*(t1-t0):.1f} ms; iters/solve {b.stats.iters_sum/b.stats.solves:.1f}")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} n={n:4d} total {1e3*t:8.2f} ms  avg {1e6*t/n:8.1f} us")
print(f"  C calls total {1e3*sum(t for _, t in acc.values()):.1f} ms")
