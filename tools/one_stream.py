import sys, time
sys.path.insert(0, '.')
import torch
from smoqyelphqmc_amd.walkers import WalkerBatch
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = sys.argv[2] if len(sys.argv) > 2 else "holstein_honeycomb_L16_Ltau128"
form = sys.argv[3] if len(sys.argv) > 3 else "sym"   # "asym": AsymFermionDetMatrix (generic kernels)
import os
# isolated per-kernel durations are taken with the CG pipeline off (full-batch launches, one after the other); SMOQY_SPLIT=0 -> the library's automatic choice
b = WalkerBatch(wl, nwalkers=nw, is_sym=form != "asym", cg_split=int(os.environ.get("SMOQY_SPLIT", "1")), device_efa=os.environ.get("SMOQY_EFA", "0") == "1", prefetch_randoms=os.environ.get("SMOQY_PREFETCH", "0") == "1")  # SMOQY_EFA=1: bench.py's sweep (device trajectory)
if os.environ.get('SMOQY_GRAPH'): b.h.call('smoqy_cg_use_graph', int(os.environ['SMOQY_GRAPH']))
if os.environ.get('SMOQY_ASYNC'): b.h.call('smoqy_hmc_async', int(os.environ['SMOQY_ASYNC']), None, None)
b.sweep(); b.sweep()
ns = int(os.environ.get("SMOQY_SWEEPS", "1"))  # timed sweeps (mean)
t0 = time.perf_counter()
for _ in range(ns): b.sweep()
b.h.call("smoqy_sync"); print("sweep ms", 1e3 * (time.perf_counter() - t0) / ns)
