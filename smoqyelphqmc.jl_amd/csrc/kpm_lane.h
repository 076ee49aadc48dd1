// Device helpers shared by the KPM kernel files (kernels_kpm.hip, kernels_kpm_wave.hip): wavefront / workgroup sums on the DPP data
// path and the "light" workgroups of the Sym Chebyshev kernels (single-term expansions, eight frequencies at a time).
#pragma once
#include "smoqy_internal.h"

namespace smoqy {

// Sum over the 64 lanes of a wavefront on the DPP data path (row shifts inside the rows of 16, then the two row broadcasts): six dependent
// steps of two v_mov_b32_dpp + one v_add_f64 each, against six ds_bpermute round trips through the LDS crossbar for the __shfl_down tree
// it replaces (round 3: the reductions sit at the end of the longest Chebyshev chain and inside every Lanczos step).  The total is
// returned in EVERY lane (read out of lane 63); the order of the additions is fixed, so results are reproducible run to run.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);  // lanes outside ROW_MASK (or with no source lane) add an exact zero
}
__device__ __forceinline__ double wsum_k(double v)
{
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds its row's sum
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wavefront's sum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ __forceinline__ double block_sum_real(double v, double *red /* >= 17 doubles */)
{
    v = wsum_k(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
    __syncthreads();  // the readers of an earlier call are done with red[]
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < nwave; ++w) t += red[w];  // every lane adds the wave sums in the same order: one value, no third barrier
    return t;
}

// Light workgroup of a Sym Chebyshev launch (see cheb_own_kernel): the `light`-th group of k.group single-term frequencies behind the
// `heavy` leading ones of system `sys` (walker w) — a scalar multiple of their vector (KPMPreconditioner.jl:398) and the Parseval
// partials of r·z.  Called by every lane of the workgroup; `lds` needs 2·8·(waves) doubles.
template <bool SPLIT>
__device__ __forceinline__ void cheb_light_workgroup(const KpmArgs &k, int sys, int w, int light, int heavy, double2 *przb, double2 *lds)
{
    const int N = k.N, Lt = k.Lt, Tn = blockDim.x, j = threadIdx.x;
    const int Lo2 = (Lt + 1) / 2;
    const int slotid = light, heavy_slots = 0;  // (ranks are counted from `heavy`)
    // ---- light workgroup: ranks [r0, r1), single-term expansions, at most GMAX of them ----
    // Three rounds instead of a loop of dependent ones: the orders and leading coefficients of all its frequencies, then all their
    // elements (a lane serves the sites j and j + Tn: N <= 2 Tn), then the stores and ONE reduction pass for all the Parseval sums.
    constexpr int GMAX = 8;
    const int r0 = heavy + (slotid - heavy_slots) * k.group, r1 = min(Lt, r0 + min(k.group, GMAX));
    const bool sys_done = k.cg[sys].done != 0;  // (k.cg is never null)
    const bool act = k.active[w] != 0;
    int omg[GMAX];
    double fg[GMAX];
    bool useg[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const int r = min(r0 + g, Lt - 1);
        const int om = (r & 1) ? Lt - 1 - (r >> 1) : (r >> 1);
        const int slot = om >= Lo2 ? Lt - om - 1 : om;  // :387
        const int n = k.order[(size_t)w * k.nslot + slot];
        const double c0 = k.coefs[((size_t)w * k.nslot + slot) * k.maxorder].x;
        omg[g] = om;
        useg[g] = r0 + g < r1 && !(k.half && om >= Lo2);
        // n > 1 here would mean the host's count of multi-term frequencies and the device's order table disagree (both come from the
        // same host vector): poison the output so that the solve fails loudly ("non-finite residual") instead of using a wrong P⁻¹
        fg[g] = !act ? k.scale : (n <= 1 ? k.scale * c0 : __builtin_nan(""));
    }
    if (sys_done) return;
    const int ia = min(j, N - 1), ib = min(j + Tn, N - 1);
    const bool oka = j < N, okb = j + Tn < N;
    double2 xa[GMAX], xb[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const double2 *v = k.v + ((size_t)omg[g] * k.nsys + sys) * N;
        xa[g] = v[ia];
        xb[g] = v[ib];
    }
    const int wave = j >> 6, lane = j & 63, nwave = (Tn + 63) >> 6;
    double *part = reinterpret_cast<double *>(lds);  // [2 GMAX][nwave] wave sums
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const double f = fg[g];
        double accr = 0.0, acci = 0.0;
        if (useg[g]) {
            double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)omg[g] * k.nsys + sys) * N;
            // the Parseval sums keep the form of the launch: per component with SPLIT (two slots per frequency), one sum otherwise;
            // per lane the terms are added in the order of the one-workgroup-per-frequency form (site j, then j + Tn)
            if (oka) {
                vo[ia] = make_double2(f * xa[g].x, f * xa[g].y);
                if constexpr (SPLIT) { accr += f * (xa[g].x * xa[g].x); acci += f * (xa[g].y * xa[g].y); }
                else accr += f * (xa[g].x * xa[g].x + xa[g].y * xa[g].y);
            }
            if (okb) {
                vo[ib] = make_double2(f * xb[g].x, f * xb[g].y);
                if constexpr (SPLIT) { accr += f * (xb[g].x * xb[g].x); acci += f * (xb[g].y * xb[g].y); }
                else accr += f * (xb[g].x * xb[g].x + xb[g].y * xb[g].y);
            }
        }
        accr = wsum_k(accr);
        acci = wsum_k(acci);
        if (lane == 0) { part[(2 * g) * nwave + wave] = accr; part[(2 * g + 1) * nwave + wave] = acci; }
    }
    __syncthreads();
    if (przb && j < 2 * GMAX) {
        const int g = j >> 1, c = j & 1;
        double t = 0.0;
        for (int q = 0; q < nwave; ++q) t += part[j * nwave + q];  // wave order, as block_sum_real
        if (r0 + g < r1 && !(k.half && (((r0 + g) & 1) ? Lt - 1 - ((r0 + g) >> 1) : ((r0 + g) >> 1)) >= Lo2)) {
            const int r = r0 + g, om = (r & 1) ? Lt - 1 - (r >> 1) : (r >> 1);
            if (SPLIT) przb[2 * om + c] = make_double2(t, 0.0);
            else if (c == 0) przb[om] = make_double2(t, 0.0);
        }
    }
}

}  // namespace smoqy
