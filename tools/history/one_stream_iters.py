"""tools/one_stream.py with the iteration count beside the time: per sweep ms, CG iterations (summed over walkers) and us per (iteration of the batch)."""
import os, sys, time
sys.path.insert(0, '.')
from smoqyelphqmc_amd.walkers import WalkerBatch
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = sys.argv[2] if len(sys.argv) > 2 else "holstein_honeycomb_L16_Ltau128"
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 4
b = WalkerBatch(wl, nwalkers=nw, is_sym=True, cg_split=int(os.environ.get("SMOQY_SPLIT", "0")), device_efa=os.environ.get("SMOQY_EFA", "1") == "1", prefetch_randoms=os.environ.get("SMOQY_PREFETCH", "1") == "1")
b.sweep(); b.sweep()
out = []
for _ in range(ns):
    i0 = b.stats.iters_sum
    t0 = time.perf_counter(); b.sweep(); b.h.call("smoqy_sync"); dt = time.perf_counter() - t0
    it = (b.stats.iters_sum - i0) / nw
    out.append(f"{1e3 * dt:.1f}ms/{it:.0f}it/{1e6 * dt / it:.1f}us")
print(" ".join(out))
