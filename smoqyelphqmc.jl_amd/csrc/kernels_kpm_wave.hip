// Sym Chebyshev apply with ONE WAVEFRONT PER CHAIN (round 4): the per-frequency recurrence of ldiv!(u', P, u)
// (src/KPMPreconditioner.jl:381-400, kpm_lmul! on B̄) for lattices whose checkerboard colours close into small groups.
//
// cheb_own_kernel gives a frequency a workgroup of N/2 lanes (two sites per lane) and pays one LDS exchange + workgroup barrier for
// every colour that is not the owned one — 1 per Chebyshev step on a chain, 4 on the square lattice — and the kernel is as long as
// its longest chain of up to a1/ϕ dependent steps (0.25 µs per step on the bond-SSH chain, 0.5 µs on the optical-SSH square lattice,
// the waves idle at s_barrier: profiles/r03_pmc_lds_iteration.txt).  Here a lane owns FOUR sites chosen so that the colours applied
// twice per step are bonds between the lane's own registers, and the others reach into a neighbouring lane of the SAME wavefront:
//
//   ring  (2 colours, N = 4·n sites, n <= 64; the bond-SSH chain L = 256 is n = 64): the two colours alternate along one cycle
//         r[0], r[1], …; lane l owns r[4l … 4l+3].  Colour 0 = (0,1), (2,3) in registers; colour 1 = (1,2) in registers,
//         site 3 with the next lane's site 0, site 0 with the previous lane's site 3 — a rotation of the wavefront by one lane
//         (DPP wave_rol / wave_ror when n = 64: register moves; ds_bpermute otherwise).
//   plaquette (4 colours, N = 4·n sites, n <= 64; the optical-SSH square lattice L = 12 is n = 36): colours 1 and 2 — the ones the
//         basis change α̃ = C₃α leaves twice in a step — close into 4-cycles s0 -c1- s1 -c2- s2 -c1- s3 -c2- s0 (the plaquettes of the
//         square lattice); a lane owns one.  Colours 1, 2 are register arithmetic; colour 0 pairs site p with site p^1 of another
//         lane, colour 3 pairs p with 3-p of another lane (the host labels the plaquettes so and VERIFIES it: api_handle.hip, wave_program):
//         four ds_bpermute pairs each.
//
// A chain therefore runs without LDS images and without a single workgroup barrier.  The two components of a frequency vector (B̄, the
// bounds and the Sym coefficients are real: cheb_own_kernel's component split) are the two wavefronts of a 128-lane workgroup, which
// never synchronise after the coefficients have been staged.  Arithmetic per site is that of cheb_own_kernel stage by stage (same
// basis change, same fused centre stage with the mate's intermediate value recomputed, same three-term recurrence); only the order
// in which the Parseval partial of r·z is summed differs.  Light workgroups (single-term frequencies) are the shared ones of kpm_lane.h.
#include "kpm_lane.h"

namespace smoqy {
namespace {

constexpr int kWaveRowsRing = 11, kWaveRowsPlaq = 28;

__device__ __forceinline__ double lin(double a, double x, double b, double y) { return a * x + b * y; }
__device__ __forceinline__ double shfl(double x, int lane) { return __shfl(x, lane, 64); }
// rotation of the whole wavefront by one lane as a DPP modifier (GFX9 wave_rol:1 / wave_ror:1): CTRL = 0x134 — lane i takes lane i + 1
// (63 takes 0); CTRL = 0x13C — lane i takes lane i - 1 (0 takes 63).  tools/dpp_probe.hip prints the two maps.
template <int CTRL>
__device__ __forceinline__ double wave_rot(double x)
{
    // every lane has a source lane under a wave rotation, so no "old" value is needed (mov_dpp: no register to clear first)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---- ring -------------------------------------------------------------------------------------------------------------------------
struct RingProg {
    int s[4];
    double d[4];
    double2 c01, c23, c12, cn, cp;  // (c̄, s̄) of the colour-0 bonds (0,1), (2,3) and of the colour-1 bonds (1,2), (3, next 0), (prev 3, 0)
    int ln, lp;
};
template <bool DPP>
__device__ __forceinline__ void ring_mates(const RingProg &P, const double *a, double &m0, double &m3)
{
    if constexpr (DPP) { m3 = wave_rot<0x134>(a[0]); m0 = wave_rot<0x13C>(a[3]); }
    else { m3 = shfl(a[0], P.ln); m0 = shfl(a[3], P.lp); }
}
// Σ_k CF[k] T_k(B̄') on the lane's four sites, n >= 2 (own_chain of kernels_kpm.hip with NCOL = 2: owned colour 0, C_L = colour 1)
template <bool DPP>
__device__ __forceinline__ void ring_chain(const RingProg &P, double *a, const double2 *CF, int n, double avg, double imag_)
{
    double m0, m3;
    // into the basis α̃ = C₁ α
    ring_mates<DPP>(P, a, m0, m3);
    {
        const double t1 = lin(P.c12.x, a[1], P.c12.y, a[2]), t2 = lin(P.c12.x, a[2], P.c12.y, a[1]);
        a[0] = lin(P.cp.x, a[0], P.cp.y, m0);
        a[3] = lin(P.cn.x, a[3], P.cn.y, m3);
        a[1] = t1; a[2] = t2;
    }
    // C₁² per site: (c² + s², 2cs) of the site's colour-1 bond
    const double qc0 = P.cp.x * P.cp.x + P.cp.y * P.cp.y, qs0 = 2.0 * P.cp.x * P.cp.y;
    const double qc12 = P.c12.x * P.c12.x + P.c12.y * P.c12.y, qs12 = 2.0 * P.c12.x * P.c12.y;
    const double qc3 = P.cn.x * P.cn.x + P.cn.y * P.cn.y, qs3 = 2.0 * P.cn.x * P.cn.y;
    // C₀ D̄ C₀ of an own bond (c, s) with the τ-means d, d' at its ends: [[c² d + s² d', c s (d + d')], [c s (d + d'), s² d + c² d']]
    const double m00 = P.c01.x * P.c01.x * P.d[0] + P.c01.y * P.c01.y * P.d[1], m11 = P.c01.y * P.c01.y * P.d[0] + P.c01.x * P.c01.x * P.d[1], m01 = P.c01.x * P.c01.y * (P.d[0] + P.d[1]);
    const double n00 = P.c23.x * P.c23.x * P.d[2] + P.c23.y * P.c23.y * P.d[3], n11 = P.c23.y * P.c23.y * P.d[2] + P.c23.x * P.c23.x * P.d[3], n01 = P.c23.x * P.c23.y * (P.d[2] + P.d[3]);
    const double imag2 = 2.0 * imag_;
    // T_k = 2 B' T_{k-1} − T_{k-2} written over T_{k-2}: the two newest vectors ping-pong between `a` and `b`, no register moves
    // (kpm_lmul!'s three-term recurrence; B' = (B̃ − avg)·imag_)
    auto apply = [&](const double *cur, double *xi) {  // xi = B̃ cur on the lane's four sites
        const double t0 = lin(m00, cur[0], m01, cur[1]), t1 = lin(m01, cur[0], m11, cur[1]);
        const double t2 = lin(n00, cur[2], n01, cur[3]), t3 = lin(n01, cur[2], n11, cur[3]);
        const double t[4] = {t0, t1, t2, t3};
        ring_mates<DPP>(P, t, m0, m3);
        xi[0] = lin(qc0, t0, qs0, m0);
        xi[1] = lin(qc12, t1, qs12, t2);
        xi[2] = lin(qc12, t2, qs12, t1);
        xi[3] = lin(qc3, t3, qs3, m3);
    };
    double b[4], acc[4], xi[4];
    {   // k = 1
        apply(a, xi);
        const double c0 = CF[0].x, c1 = CF[1].x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            b[p] = imag_ * (xi[p] - avg * a[p]);
            acc[p] = c0 * a[p] + c1 * b[p];
        }
    }
    auto step = [&](const double *cur, double *old, double ck) {
        apply(cur, xi);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            old[p] = imag2 * (xi[p] - avg * cur[p]) - old[p];
            acc[p] += ck * old[p];
        }
    };
    int kk = 2;
    for (; kk + 1 < n; kk += 2) {
        step(b, a, CF[kk].x);
        step(a, b, CF[kk + 1].x);
    }
    if (kk < n) step(b, a, CF[kk].x);
    // back to the original basis: C₁⁻¹ on the accumulated sum
#pragma unroll
    for (int p = 0; p < 4; ++p) a[p] = acc[p];
    ring_mates<DPP>(P, a, m0, m3);
    const double i0 = 1.0 / (P.cp.x * P.cp.x - P.cp.y * P.cp.y), i12 = 1.0 / (P.c12.x * P.c12.x - P.c12.y * P.c12.y), i3 = 1.0 / (P.cn.x * P.cn.x - P.cn.y * P.cn.y);
    a[0] = i0 * (P.cp.x * acc[0] - P.cp.y * m0);
    a[1] = i12 * (P.c12.x * acc[1] - P.c12.y * acc[2]);
    a[2] = i12 * (P.c12.x * acc[2] - P.c12.y * acc[1]);
    a[3] = i3 * (P.cn.x * acc[3] - P.cn.y * m3);
}

// ---- plaquette --------------------------------------------------------------------------------------------------------------------
struct PlaqProg {
    int s[4];
    double d[4], dm[4];             // τ-means of exp(-ΔτV) at the own sites and at their colour-0 mates
    double2 c1a, c1b, c2a, c2b;     // colour 1: (0,1), (2,3); colour 2: (1,2), (3,0)
    double2 c0[4], c3[4];           // colour 0 / colour 3 bond of each own site
    int l0[4], l3[4];               // lanes holding the colour-0 / colour-3 mates (at positions p^1 / 3-p)
};
#define PLAQ_PAIR(v_, cs_, i_, j_)                                              \
    {                                                                           \
        const double t_ = lin(cs_.x, v_[i_], cs_.y, v_[j_]);                    \
        v_[j_] = lin(cs_.x, v_[j_], cs_.y, v_[i_]);                             \
        v_[i_] = t_;                                                            \
    }
__device__ __forceinline__ void plaq_chain(const PlaqProg &P, double *a, const double2 *CF, int n, double avg, double imag_)
{
    double m[4];
    // into the basis α̃ = C₃ α
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = shfl(a[3 - p], P.l3[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) a[p] = lin(P.c3[p].x, a[p], P.c3[p].y, m[p]);
    double qc[4], qs[4], e0[4], e1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        qc[p] = P.c3[p].x * P.c3[p].x + P.c3[p].y * P.c3[p].y; qs[p] = 2.0 * P.c3[p].x * P.c3[p].y;
        // C₀ D̄ C₀ on an own site whose colour-0 mate lives elsewhere: a' = (c² d + s² d_mate) a + c s (d + d_mate) mate
        e0[p] = P.c0[p].x * P.c0[p].x * P.d[p] + P.c0[p].y * P.c0[p].y * P.dm[p];
        e1[p] = P.c0[p].x * P.c0[p].y * (P.d[p] + P.dm[p]);
    }
    const double imag2 = 2.0 * imag_;
    // the two newest vectors of the three-term recurrence ping-pong between `a` and `b` (see ring_chain)
    auto apply = [&](const double *cur, double *xi) {  // xi = B̃ cur = C₃² C₂ C₁ (C₀ D̄ C₀) C₁ C₂ cur
        double t[4] = {cur[0], cur[1], cur[2], cur[3]};
        PLAQ_PAIR(t, P.c2a, 1, 2) PLAQ_PAIR(t, P.c2b, 3, 0)   // C₂
        PLAQ_PAIR(t, P.c1a, 0, 1) PLAQ_PAIR(t, P.c1b, 2, 3)   // C₁
#pragma unroll
        for (int p = 0; p < 4; ++p) m[p] = shfl(t[p ^ 1], P.l0[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) t[p] = lin(e0[p], t[p], e1[p], m[p]);   // C₀ D̄ C₀, the mate's part folded into (e0, e1)
        PLAQ_PAIR(t, P.c1a, 0, 1) PLAQ_PAIR(t, P.c1b, 2, 3)   // C₁
        PLAQ_PAIR(t, P.c2a, 1, 2) PLAQ_PAIR(t, P.c2b, 3, 0)   // C₂
#pragma unroll
        for (int p = 0; p < 4; ++p) m[p] = shfl(t[3 - p], P.l3[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) xi[p] = lin(qc[p], t[p], qs[p], m[p]);   // C₃²
    };
    double b[4], acc[4], xi[4];
    {   // k = 1
        apply(a, xi);
        const double c0 = CF[0].x, c1 = CF[1].x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            b[p] = imag_ * (xi[p] - avg * a[p]);
            acc[p] = c0 * a[p] + c1 * b[p];
        }
    }
    auto step = [&](const double *cur, double *old, double ck) {
        apply(cur, xi);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            old[p] = imag2 * (xi[p] - avg * cur[p]) - old[p];
            acc[p] += ck * old[p];
        }
    };
    int kk = 2;
    for (; kk + 1 < n; kk += 2) {
        step(b, a, CF[kk].x);
        step(a, b, CF[kk + 1].x);
    }
    if (kk < n) step(b, a, CF[kk].x);
#pragma unroll
    for (int p = 0; p < 4; ++p) a[p] = acc[p];
#pragma unroll
    for (int p = 0; p < 4; ++p) m[p] = shfl(a[3 - p], P.l3[p]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double idet = 1.0 / (P.c3[p].x * P.c3[p].x - P.c3[p].y * P.c3[p].y);
        a[p] = idet * (P.c3[p].x * acc[p] - P.c3[p].y * m[p]);
    }
}
#undef PLAQ_PAIR

// KIND 1: ring through ds_bpermute, 2: ring with n = 64 lanes (DPP wave rotations), 3: plaquette
template <int KIND>
__global__ void __launch_bounds__(128) cheb_wave_kernel(KpmArgs k, KpmGeom kg)
{
    extern __shared__ double2 lds[];
    const int N = k.N, Lt = k.Lt, j = threadIdx.x;
    const int ncnt_ = k.sys_count > 0 ? k.sys_count : k.nsys;
    int sys = k.sys_first + blockIdx.x % ncnt_, slotid = blockIdx.x / ncnt_;
    if (k.xcd_map && (ncnt_ & 7) == 0) {  // XCD x takes the contiguous share [x·n/8, (x+1)·n/8) of the systems, as in every kernel of the iteration
        const int per_ = ncnt_ >> 3, q_ = blockIdx.x >> 3;
        sys = k.sys_first + (blockIdx.x & 7) * per_ + q_ % per_;
        slotid = q_ / per_;
    }
    const int heavy = min(k.heavy, Lt);
    const int w = sys / k.nrhs;
    const int Lo2 = (Lt + 1) / 2;
    double2 *przb = k.part_rz ? k.part_rz + (size_t)sys * k.rz_stride : nullptr;
    if (slotid >= heavy) {
        cheb_light_workgroup<true>(k, sys, w, slotid - heavy, heavy, przb, lds);
        return;
    }
    // ---- heavy workgroup: one frequency; wavefront 0 runs the real part of its vector, wavefront 1 the imaginary part ----
    double2 *CF = lds;
    const int comp = j >> 6, lane = j & 63, rank = slotid;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);
    if (k.half && om >= Lo2) return;
    const int slot = om >= Lo2 ? Lt - om - 1 : om;  // :387
    // round 1 of loads: everything addressed by the block and thread index alone
    const bool sys_done = k.cg[sys].done != 0;
    const bool act = k.active[w] != 0;
    const int n_raw = k.order[(size_t)w * k.nslot + slot];
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    constexpr int ROWS = KIND == 3 ? kWaveRowsPlaq : kWaveRowsRing;
    int tab[ROWS];
    const bool on = lane < kg.wave_lanes;
    const int lq = on ? lane : 0;  // lanes past the table load entry 0 and never store
#pragma unroll
    for (int r = 0; r < ROWS; ++r) tab[r] = kg.wave[r * 64 + lq];
    asm volatile("" ::: "memory");
    if (sys_done) return;
    const double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)om * k.nsys + sys) * N;
    double2 *prz = przb ? przb + 2 * om + comp : nullptr;
    const int n = act ? n_raw : 1;
    const double2 *coefs = k.coefs + ((size_t)w * k.nslot + slot) * k.maxorder;
    if (n <= 1) {
        // single-term expansion (:398) that the host's count of chain-carrying frequencies put in front of the light ones (the count is an
        // upper bound; it may lag the device's table by an update).  Run it EXACTLY as a light workgroup would — same lanes, same order of
        // the Parseval sum — so that r·z, and with it every iterate of the solve, does not depend on the host's count.
        KpmArgs k1 = k;
        k1.group = 1;
        cheb_light_workgroup<true>(k1, sys, w, 0, rank, przb, lds);
        return;
    }
    const double avg = 0.5 * (emax + emin), imag_ = 1.0 / (0.5 * (emax - emin));
    // round 2: the gathers addressed through round 1
    const double *dbar = k.dbar + (size_t)w * N;
    const double2 *pcs = kg.pcs + (size_t)w * kg.ptotal;
    double a[4], v0[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double2 x = v[tab[p]];
        a[p] = on ? (comp ? x.y : x.x) : 0.0;
        v0[p] = a[p];
    }
    for (int i = j; i < n; i += 128) CF[i] = coefs[i];  // coefficients in LDS: no global load inside the chain
    if constexpr (KIND == 3) {
        PlaqProg P;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            P.s[p] = tab[p];
            P.d[p] = dbar[tab[p]];
            P.dm[p] = dbar[tab[24 + p]];
            P.c0[p] = pcs[tab[8 + p]];
            P.c3[p] = pcs[tab[12 + p]];
            P.l0[p] = tab[16 + p];
            P.l3[p] = tab[20 + p];
        }
        P.c1a = pcs[tab[4]]; P.c1b = pcs[tab[5]]; P.c2a = pcs[tab[6]]; P.c2b = pcs[tab[7]];
        __syncthreads();  // CF visible; the only barrier of the workgroup
        plaq_chain(P, a, CF, n, avg, imag_);
    } else {
        RingProg P;
#pragma unroll
        for (int p = 0; p < 4; ++p) { P.s[p] = tab[p]; P.d[p] = dbar[tab[p]]; }
        P.c01 = pcs[tab[4]]; P.c23 = pcs[tab[5]]; P.c12 = pcs[tab[6]]; P.cn = pcs[tab[7]]; P.cp = pcs[tab[8]];
        P.ln = tab[9]; P.lp = tab[10];
        __syncthreads();
        ring_chain<KIND == 2>(P, a, CF, n, avg, imag_);
    }
    double acc = 0.0;
    if (on) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double z = k.scale * a[p];
            if (comp) vo[tab[p]].y = z; else vo[tab[p]].x = z;
            acc += v0[p] * z;  // Re conj(r)·z: this wavefront's component (the imaginary part is rounding noise around an exact zero and is not formed)
        }
    }
    acc = wsum_k(acc);
    if (prz && lane == 0) *prz = make_double2(acc, 0.0);
}

}  // namespace

bool cheb_wave_supported(const KpmArgs &k, const KpmGeom &kg)
{
    static const int env = tuning_env(kTuneChebWave);
    return env != 0 && kg.wave_kind != 0 && kg.wave && k.is_sym && k.N <= 256 && cheb_split_active(k, kg);
}

void launch_cheb_wave(hipStream_t st, const KpmArgs &k, const KpmGeom &kg)
{
    const int ncnt = k.sys_count > 0 ? k.sys_count : k.nsys;
    KpmArgs kk = k;
    kk.group = std::min(8, k.group);
    if (kk.group <= 1) { kk.group = 1; kk.heavy = k.Lt; }
    kk.heavy = std::max(0, std::min(k.Lt, kk.heavy));
    const int nlight = (k.Lt - kk.heavy + kk.group - 1) / kk.group;
    const size_t lds = sizeof(double2) * (size_t)std::max(k.maxorder, 32);
    const dim3 grid((unsigned)((kk.heavy + nlight) * ncnt)), block(128);
    if (kg.wave_kind == 2) hipLaunchKernelGGL((cheb_wave_kernel<3>), grid, block, lds, st, kk, kg);
    else if (kg.wave_lanes == 64 && tuning_env(kTuneChebWave) != 2) hipLaunchKernelGGL((cheb_wave_kernel<2>), grid, block, lds, st, kk, kg);
    else hipLaunchKernelGGL((cheb_wave_kernel<1>), grid, block, lds, st, kk, kg);
}

}  // namespace smoqy
