"""The normal deviates of the native member threads (smoqy_team_bench_sweeps, csrc/team.hip: xoshiro256++ and a 128-layer ziggurat), through
smoqy_bench_randn — host code, no GPU.  The members' random numbers stand in for the randn! calls of src/PFFCalculator.jl:67 and
src/KPMPreconditioner.jl:634; a generator with the wrong distribution would change what the measured sweeps solve."""
import math

import numpy as np
from scipy import stats

from smoqyelphqmc_amd import _lib as L


def _draw(n, seed, scale=1.0):
    a = np.empty(n)
    assert L.load().smoqy_bench_randn(L.ptr(a), n, seed, scale) == 0
    return a


def test_moments_tail_and_kolmogorov_smirnov():
    n = 2_000_000
    a = _draw(n, 12345)
    assert abs(a.mean()) < 4 / math.sqrt(n)
    assert abs(a.var() - 1) < 4 * math.sqrt(2 / n)
    assert abs(((a - a.mean()) ** 4).mean() / a.var() ** 2 - 3) < 4 * math.sqrt(24 / n)
    r = 3.442619855899  # the ziggurat's last layer: beyond it the tail sampler takes over
    p_tail = math.erfc(r / math.sqrt(2))
    assert abs((np.abs(a) > r).mean() - p_tail) < 5 * math.sqrt(p_tail / n)
    assert abs((a > 0).mean() - 0.5) < 4 * 0.5 / math.sqrt(n)
    assert stats.kstest(a[: n // 2], "norm").pvalue > 1e-3
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 4 / math.sqrt(n)


def test_streams_scale_and_arguments():
    a, b = _draw(100_000, 7), _draw(100_000, 8)
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.02                     # different seeds, different streams
    np.testing.assert_array_equal(a, _draw(100_000, 7))            # the same seed, the same stream
    np.testing.assert_allclose(_draw(1000, 7, math.sqrt(0.5)), math.sqrt(0.5) * a[:1000], rtol=1e-15)
    assert _draw(0, 1).size == 0
    assert L.load().smoqy_bench_randn(None, 4, 1, 1.0) != 0
