"""Static register / scratch / LDS budgets of the gfx950 kernels, read from the built code objects (tools/kernel_resources.py; no GPU).

What the design rests on and a profile cannot show when a later edit tips a kernel over an allocation step: the four kernels of a CG
iteration (mul_MtM! src/FermionDetMatrix.jl:329-340, the FourierTransformer pair src/FourierTransformer.jl:39-64 with the CG updates
src/IterativeSolvers/ConjugateGradient.jl:219-245, kpm_lmul! under src/KPMPreconditioner.jl:381-400) run without scratch, and each keeps
the wavefronts per SIMD that DESIGN.md / docs/DESIGN_LOG.md quote for it.  profiles/r04_kernel_resources.txt is the tool's table at HEAD."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
KR = importlib.util.module_from_spec(spec)
spec.loader.exec_module(KR)


@pytest.fixture(scope="module")
def kernels():
    if not any(f.endswith(".o") for f in os.listdir(KR.CSRC)):
        pytest.skip("object files not built (run __graft_entry__.build())")
    ks = KR.collect()
    assert len(ks) > 250        # every instantiation of every family is in the library
    return {k["kernel"]: k for k in ks}


def test_occupancy_rule_of_the_unified_register_file():
    # MI355X_MICROARCH.md: 512 registers per SIMD lane, granule 8: <= 64 -> 8 wavefronts, 104-128 -> 4, 176-256 -> 2, 264-512 -> 1
    assert [KR.waves_per_simd(v) for v in (40, 64, 72, 96, 128, 129, 168, 176, 256, 257, 258, 512)] == [8, 8, 7, 5, 4, 3, 3, 2, 2, 1, 1, 1]


# kernel (as launched by bench.py on the five BASELINE configs, profiles/r04_*bench_kernel_stats.txt) -> wavefronts per SIMD it must keep
HOT = {
    "fdm_stream_kernel<3, false, true, 256>": 4,      # headline MᵀM, 16 systems per launch (128 VGPRs)
    "fdm_stream_kernel<2, true, true, 256>": 3,       # bond-SSH chain
    "fdm_own_stream_kernel<3, 256>": 3,               # honeycomb L = 8, 64 systems per launch
    "fdm_own_kernel<3, 0, 2, 256>": 4,                # one walker
    "fdm_wave_kernel<wave_desc::PlaqD, 2, false, 1>": 2,   # optical-SSH square
    "cheb_own_kernel<3, true, 3>": 5, "cheb_own_kernel<3, true, 1>": 5,
    "cheb_wave_kernel<2>": 4, "cheb_wave_kernel<3>": 3,
    "tfft_kernel<2, false, true>": 6, "tfft_kernel<3, false, true>": 6, "tfft_kernel<2, true, false>": 7, "tfft_kernel<3, true, false>": 7,
    "tfft_rb_kernel<2, 5, 8, 8>": 4, "tfft_rb_kernel<3, 5, 8, 8>": 4, "tfft_rb_kernel<2, 4, 5, 16>": 5, "tfft_rb_kernel<3, 4, 5, 16>": 5,
    "tfft_rb_kernel<2, 5, 4, 16>": 5, "tfft_rb_kernel<3, 5, 4, 16>": 4,
    "lanczos_own_kernel<3, 3>": 5, "dmdx_fast_kernel<3>": 8, "efa_kernel": 4, "cg_finish_kernel": 8,
}


@pytest.mark.parametrize("name", sorted(HOT))
def test_hot_kernels_have_no_scratch_and_keep_their_occupancy(kernels, name):
    k = kernels[name]
    assert k.get("scratch", 0) == 0 and k.get("vgpr_spill", 0) == 0, k
    assert k["waves_per_simd"] >= HOT[name], k
    assert not str(k.get("dyn_stack", "false")).lower().startswith("t"), k


def test_honeycomb_block_program_and_its_two_wave_twin(kernels):
    """fdm_wave_kernel<HoneyD>: eight complex sites per lane.  As launched by default it takes 256 VGPRs + 2 AGPRs -> 264 allocated -> ONE
    wavefront per SIMD (what profiles/r04_pmc_explore_sq_mtm_b128_wave.txt shows: SQ_WAVE_CYCLES ~ SQ_BUSY_CU_CYCLES); the twin of
    SMOQY_FDM_WAVE_OCC=2 (centre coefficients formed inside each propagate instead of held) must fit 256 without scratch or AGPRs."""
    one = kernels["fdm_wave_kernel<wave_desc::HoneyD, 0, false, 1>"]
    two = kernels["fdm_wave_kernel<wave_desc::HoneyD, 0, false, 2>"]
    assert one["scratch"] == 0 and one["waves_per_simd"] == 1 and 256 < one["vgpr"] <= 264, one
    assert two["waves_per_simd"] == 2 and two["vgpr"] <= 256 and two["scratch"] == 0 and two["agpr"] == 0 and two.get("vgpr_spill", 0) == 0, two


def test_kernels_with_scratch_are_only_the_capped_1024_lane_and_wide_colour_forms(kernels):
    """Scratch is tolerated where a 1024-lane workgroup caps the kernel at 128 VGPRs (big-lattice instantiations no BASELINE config
    launches), in the five- and six-colour Chebyshev forms and the complex-hopping forms — nowhere else."""
    allowed = re.compile(r"(, 1024>$)|(^fdm_fast_kernel<[34], \d, true, (true|false)>$)|(^cheb_own(_asym)?_kernel<[56])|(^cheb_fast_kernel<true, 0, true>$)"
                         )
    bad = [n for n, k in kernels.items() if (k.get("scratch", 0) or k.get("vgpr_spill", 0)) and not allowed.search(n)]
    assert not bad, bad
