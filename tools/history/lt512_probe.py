import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import test_gpu_edge_cases as T
from test_gpu_edge_cases import *
for Lt in (320, 512):
    N=24
    h, o, nt, colors = make(lat.chain_neighbor_table(N), Lt, N, True, seed=3, nrhs=3, vscale=0.5)
    v = rand(Lt, N, 3, 4)
    rv = np.random.default_rng(5).standard_normal(N)
    P = orc.OracleKPM(o[0]); P.update(rv)
    h.call("smoqy_precond_update", 0, L.ptr(rv))
    res={}
    for form in (0,1):
        h.call("smoqy_tfft_form", form)
        res[form]=solve(h, v, 1e-10, 5000, 1)
    xo, ito, _ = o[0].cg_solve(v[:, :, 0], precond=P, tol=1e-10, maxiter=5000)
    print(Lt, res[0][1], res[1][1], ito, relerr(res[1][0][:,:,0], xo), relerr(res[0][0][:,:,0], xo))
