"""Launch only the stream-copy kernel (calibration for rocprofv3 --pmc passes): `python tools/copy_only.py [MiB] [reps]`."""
import sys
sys.path.insert(0, '.')
import ctypes as C
import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lat = sq.lattice
m = lat.holstein_honeycomb(4, 40)
nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
h = L.Handle(40, 32, nt, colors, True, 1, 1, -1)
ms = C.c_double(0.0)
h.call("smoqy_bench_copy", C.c_size_t(mib << 20), reps, C.byref(ms))
print(f"copy of {mib} MiB: {2 * (mib << 20) / (ms.value / reps * 1e-3) / 1e9:.0f} GB/s")
