"""The whole sweep bench.py times, device against oracle, number for number: one WalkerBatch sweep (two global moves + hmc_update! with
the EFA leapfrog on the device) and oracle/sweep.py's restatement of it on the same walkers and the same PCG64 streams — iteration
counts per solve, the final action and the trajectory's ΔH.  PARITY UNPINNED like everything that rests on the oracle (the leapfrog
doubly so: oracle/efa.py header)."""
import numpy as np
import pytest

from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle.sweep import OracleWalker

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,nw,prefetch", [("holstein_honeycomb_L4_Ltau40", 3, True), ("holstein_honeycomb_L4_Ltau40", 2, False), ("holstein_honeycomb_L8_Ltau80", 2, True),
                                              ("bssh_chain_L256_Ltau200_alpha0p2", 2, True), ("ossh_square_L12_Ltau100_alpha0p2", 2, True)])
def test_sweep_matches_the_oracle_solve_by_solve(name, nw, prefetch):
    b = WalkerBatch(name, nwalkers=nw, walker0=5, device_efa=True, prefetch_randoms=prefetch)
    b.iter_log = []
    ws = [OracleWalker(name, walker=5 + w) for w in range(nw)]
    for sweep in range(2):                                   # the second sweep starts from the restored fields and the advanced generators
        del b.iter_log[:]
        last = b.sweep()
        dH = b.dH.copy()
        got = np.stack(b.iter_log)                           # (27, nw)
        for w in range(nw):
            r = ws[w].sweep()
            want = np.asarray(r["iters"])
            assert got.shape[0] == want.shape[0] == 27
            assert np.abs(got[:, w] - want).max() <= 1, (sweep, w, got[:, w], want)   # another summation order: at most one step apart
            assert abs(last[0][w] - r["action"]) < 1e-7 * abs(r["action"])             # final action, solved to tol = 1e-10
            # ΔH is a difference of O(10³) energies along 24 force solves at sqrt(tol) = 1e-5: the two drivers agree far below its size
            assert abs(dH[w] - r["dH"]) < 1e-5 * max(1.0, abs(r["action"])) , (dH[w], r["dH"])
    b.h.close()
