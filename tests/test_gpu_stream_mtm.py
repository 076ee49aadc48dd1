"""Streaming form of the fused MᵀM (fdm_stream_kernel, smoqy_matvec_stream) against the chunked kernels and the oracle.

mul_MtM! = mul_Mt!(mul_M!) (src/FermionDetMatrix.jl:329-340).  The streaming kernel runs the same stage order per site, so its output must be
BIT-IDENTICAL to the chunked register-resident kernel's; its p·Ap partial is |M p|² instead of p·(MᵀM p) (equal up to rounding), checked
through a CG solve (same solution, iteration count within one step)."""
import ctypes as C

import numpy as np
import pytest

import smoqyelphqmc_amd as sq
from smoqyelphqmc_amd import _lib as L
from smoqyelphqmc_amd.walkers import WalkerBatch
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
lat = sq.lattice


def relerr(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _handle(m, nw, walkers=None):
    nt, perm, colors = lat.checkerboard_decomposition(m.fpi.neighbor_table)
    h = L.Handle(m.fpi.Ltau, m.fpi.N, nt, colors, True, nw, 1)
    for w in range(nw):
        mw = m if walkers is None else walkers[w]
        h.call("smoqy_update_from_path_integral", w, L.ptr(mw.fpi.V), L.ptr(mw.fpi.t), L.ptr(perm), C.c_double(mw.fpi.dtau))
    h.call("smoqy_matvec_wave", 0)  # this file is about the WORKGROUP kernels: the one-wavefront-per-run kernel (tests/test_gpu_wave_mtm.py) stays out
    return h, nt, perm


@pytest.mark.parametrize("case", ["honeycomb_L4_Lt12", "honeycomb_L4_Lt13", "chain_L24_Lt9", "chain_L24_Lt16_const", "square_L6_Lt12_const", "square_L6_Lt12_ssh", "honeycomb_L3_Lt5"])
@pytest.mark.parametrize("R", [2, 4, 6, 64])
def test_stream_equals_chunked_and_oracle(case, R):
    """Odd and even Lτ, runs that do not divide Lτ, a run longer than Lτ, 1 / 2 / 3 / 4 colours, τ-dependent hoppings on two colours."""
    if case == "honeycomb_L4_Lt12":
        ms = [lat.holstein_honeycomb(4, 12, walker=w) for w in range(3)]
    elif case == "honeycomb_L4_Lt13":
        ms = [lat.holstein_honeycomb(4, 13, walker=w) for w in range(3)]
    elif case == "honeycomb_L3_Lt5":
        ms = [lat.holstein_honeycomb(3, 5, walker=w) for w in range(2)]
    elif case == "chain_L24_Lt9":
        ms = [lat.bssh_chain(24, 9, walker=w) for w in range(3)]       # two colours, hoppings depend on τ
    elif case == "chain_L24_Lt16_const":
        ms = [lat.bssh_chain(24, 16, alpha=0.0, walker=w) for w in range(2)]
    elif case == "square_L6_Lt12_ssh":
        ms = [lat.ossh_square(6, 12, walker=w) for w in range(2)]              # four colours, hoppings depend on τ (256-lane instantiation)
    else:
        ms = [lat.ossh_square(6, 12, alpha=0.0, walker=w) for w in range(2)]  # four colours, constant hoppings
    nw = len(ms)
    h, nt, perm = _handle(ms[0], nw, ms)
    Lt, N = ms[0].fpi.Ltau, ms[0].fpi.N
    g = np.random.default_rng(5)
    v = np.asfortranarray(g.standard_normal((Lt, N, nw)) + 1j * g.standard_normal((Lt, N, nw)))
    a, b, c = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    for Tc in (1, 2):
        h.call("smoqy_set_tau_chunk", Tc)
        h.call("smoqy_matvec_stream", 0)
        h.call("smoqy_matvec_v", L.OP_MTM, b, a)
        chunked = h.vec_download(b)
        h.call("smoqy_matvec_stream", -1)  # automatic: chunked / owner-computes at this size
        h.call("smoqy_matvec_v", L.OP_MTM, c, a)
        assert np.array_equal(h.vec_download(c), chunked)
        h.call("smoqy_matvec_stream", R)
        h.call("smoqy_matvec_v", L.OP_MTM, c, a)
        stream = h.vec_download(c)
        # a few systems per launch: the chunked side is the owner-computes kernel (another operation order): equal to rounding;
        # bit-identity against fdm_fast_kernel is asserted at the benchmarked lattice below
        assert relerr(stream, chunked) < 1e-14, (case, R, Tc, np.abs(stream - chunked).max())
    for w, m in enumerate(ms):
        expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, perm, m.fpi.dtau, True)
        o = orc.OracleFDM(nt, expV, ch, sh, True)
        assert relerr(stream[:, :, w], o.mul_MtM(v[:, :, w])) < 1e-13
    # the CG (twiddled operator, p·Ap partials from the streaming kernel) reaches the same solution
    h.call("smoqy_set_tau_chunk", 0)
    rv = np.ascontiguousarray(g.standard_normal((nw, N)))
    h.call("smoqy_precond_update_all", L.ptr(rv))
    its = {}
    sols = {}
    for r in (0, R):
        h.call("smoqy_matvec_stream", r)
        h.vec_upload(b, v)
        it, eps = np.zeros(nw, dtype=np.int32), np.zeros(nw)
        h.call("smoqy_cg_solve_v", b, b, C.c_double(1e-10), 10000, 1, L.ptr(it), L.ptr(eps))
        its[r], sols[r] = it.copy(), h.vec_download(b)
        assert np.all(eps < 1e-10)
    assert np.all(np.abs(its[0] - its[R]) <= 1), (its[0], its[R])
    assert relerr(sols[R], sols[0]) < 1e-8
    h.close()


@pytest.mark.parametrize("nw,R", [(16, 4), (16, 16), (64, 16)])
def test_stream_at_the_benchmarked_lattice(nw, R):
    batch = WalkerBatch("holstein_honeycomb_L16_Ltau128", nwalkers=nw)
    h = batch.h
    h.call("smoqy_matvec_wave", 0)
    g = np.random.default_rng(6)
    v = np.asfortranarray(g.standard_normal((batch.Lt, batch.N, nw)) + 1j * g.standard_normal((batch.Lt, batch.N, nw)))
    a, b, c = h.vec_alloc(), h.vec_alloc(), h.vec_alloc()
    h.vec_upload(a, v)
    h.call("smoqy_matvec_stream", 0)
    h.call("smoqy_matvec_v", L.OP_MTM, b, a)
    h.call("smoqy_matvec_stream", R)
    h.call("smoqy_matvec_v", L.OP_MTM, c, a)
    chunked, stream = h.vec_download(b), h.vec_download(c)
    assert relerr(stream, chunked) < 1e-14  # same stage order per site; the compiler contracts the two kernels' arithmetic differently
    w = nw - 1
    m = batch.models[w]
    expV, ch, sh = orc.update_fields(m.fpi.V, m.fpi.t, batch.perm, m.fpi.dtau, True)
    o = orc.OracleFDM(batch.nt, expV, ch, sh, True)
    assert relerr(stream[:, :, w], o.mul_MtM(v[:, :, w])) < 1e-13
    h.close()
