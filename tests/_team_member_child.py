"""Child process of tests/test_gpu_team.py: ONE rank of the reference's one-walker-per-rank model, joining a team another process serves.
It never touches the GPU.  argv: info-json, walker index, mode ("parity" out.npz | "hmc" out.npz | "ge" out.npz | "sweeps" n)."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smoqyelphqmc_amd import _lib as L  # noqa: E402
from smoqyelphqmc_amd.walkers import RemoteMember  # noqa: E402

info, w, mode = json.loads(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
m = RemoteMember(info, w, seed=info.get("seed", 0))
if mode == "parity":
    g = np.random.default_rng([info["seed"], w])  # the parent draws the same numbers
    R = np.asfortranarray((g.standard_normal((m.Lt, m.N)) + 1j * g.standard_normal((m.Lt, m.N))) * np.sqrt(0.5))
    x = m.x.copy()
    x[:, : m.free] += 0.05 * g.standard_normal((m.Lt, m.free))
    rv = g.standard_normal(m.N)
    rr = C.c_double(0.0)
    m._sample_call(L.ptr(R), C.byref(rr))
    s, e, i = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
    d = np.zeros((m.Lt, m.Nph))
    m._step_call(L.ptr(x), L.ptr(rv), C.c_double(1e-10), 10000, 1, C.byref(s), C.byref(i), C.byref(e), L.ptr(d))
    np.savez(sys.argv[4], rr=rr.value, sf=s.value, it=i.value, eps=e.value, dS=d)
elif mode == "hmc":
    from smoqyelphqmc_amd import lattice as lat  # noqa: E402

    m.rng = np.random.Generator(np.random.PCG64(lat.SEED0 + 7919 * w + 1))  # walker w's stream, as WalkerBatch seeds it
    dH, x_new = m.hmc_update()
    m.hmc_finish(w % 2 == 0, x_new)
    np.savez(sys.argv[4], dH=dH, x_new=x_new)
elif mode == "ge":
    from smoqyelphqmc_amd import lattice as lat  # noqa: E402

    m.rng = np.random.Generator(np.random.PCG64(lat.SEED0 + 7919 * w + 1))  # walker w's stream, as WalkerBatch seeds it
    G, it = m.measure_greens(orbitals=(1, 2))
    np.savez(sys.argv[4], G=G, it=it)
else:
    for _ in range(int(sys.argv[4])):
        m.sweep()
    print(json.dumps({"w": w, "solves": m.solves, "iters": m.iters_sum}))
m.close()
