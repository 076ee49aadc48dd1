// KPM preconditioner kernels for gfx950: per-frequency Chebyshev recurrence in the
// tau-averaged propagator B̄ and the Lanczos eigen-bound estimate.
//
// Reference semantics: ldiv!(u', P, u) src/KPMPreconditioner.jl:355-414 (Sym), :488-550 (Asym);
// calculate_bounds! :625-658; update_B̄! :604-621; B̄ = Γ̄ D̄ Γ̄ᴴ (Sym) / D̄ Γ̄ (Asym) as in
// JDQMCFramework's Sym/AsymChkbrdPropagator; kpm_lmul!/lanczos! restated from SmoQyKPMCore
// (three-term Chebyshev recurrence on B̄ rescaled to [-1,1]; plain Lanczos) — third-party source
// absent, see DESIGN.md.
//
// Mapping.  After the tau-FFT the device layout v[ω][s][i] already has one frequency of one
// system as a contiguous N-vector, so there is no transpose (the reference needs two,
// :378/:403).  One workgroup owns one (ω, system); the chain of up to ~a1/φ dependent B̄ applies
// is pure latency, so the kernel is organised around the number of barrier-separated stages:
//   * every colour's bond list is padded with identity "self bonds" so that it covers all N
//     sites; a lane owns one (padded) bond per colour and keeps its site pair and the (c̄, s̄)
//     pair of every colour in registers for the whole chain — no memory traffic besides LDS;
//   * Sym: the diagonal D̄ is folded into the first colour's stage (C₁ D̄ C₁ in registers), and
//     the recurrence runs in the basis α̃ = C_L α, where B̃ = C_L B̄ C_L⁻¹ = C_L² C_{L-1}…C₁D̄C₁…C_{L-1}
//     merges the two applications of the last colour: 2L-2 stages per step instead of 2L+1;
//   * the three-term recurrence and the running sum live in the registers of the lane that owns
//     the last colour's bond and are fused into that colour's stage.
// Workgroups are issued heaviest expansion order first.  A generic LDS-table kernel remains as
// the fallback for decompositions with more than kMaxColours colours or more padded bonds per
// colour than 1024.
#include "smoqy_internal.h"
#include "kpm_lane.h"

#include <cstdlib>

namespace smoqy {

__device__ __forceinline__ double2 cmulk(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 axpy2(double a, double2 x, double2 y) { return make_double2(a * x.x + y.x, a * x.y + y.y); }
__device__ __forceinline__ double2 lin2(double a, double2 x, double b, double2 y) { return make_double2(a * x.x + b * y.x, a * x.y + b * y.y); }

// ---------------------------------------------------------------------------------------------
// update_B̄! (:604-621) for every walker in one launch, plus the padded (c̄, s̄) table
// ---------------------------------------------------------------------------------------------
__global__ void tau_means_kernel(const double *__restrict__ expV, const double *__restrict__ ch, const double *__restrict__ sh, double *dbar, double *cbar, double *sbar, double2 *pcs,
                                 const int *__restrict__ psrc, int ptotal, int Lt, int N, int Nh, const double *__restrict__ shi, double *sbari, double *pcsi)
{
    // 64 outputs x 4 tau-groups per workgroup: lanes run over sites/bonds (coalesced), each
    // thread sums every 4th slice with 8 loads in flight, then one LDS hop across the groups
    __shared__ double2 part[4][64];
    __shared__ double parti[4][64];
    const int w = blockIdx.y;
    const int j = blockIdx.x * 64 + threadIdx.x, ty = threadIdx.y;
    expV += (size_t)w * Lt * N; ch += (size_t)w * Lt * Nh; sh += (size_t)w * Lt * Nh;
    if (shi) shi += (size_t)w * Lt * Nh;
    double a = 0, b = 0, bi = 0;
    int h = -1;
    if (j < N) {
#pragma unroll 8
        for (int l = ty; l < Lt; l += 4) a += expV[(size_t)l * N + j];
    } else if (j < N + ptotal) {
        h = psrc[j - N];
        if (h >= 0) {
#pragma unroll 8
            for (int l = ty; l < Lt; l += 4) { a += ch[(size_t)l * Nh + h]; b += sh[(size_t)l * Nh + h]; }
            if (shi)
                for (int l = ty; l < Lt; l += 4) bi += shi[(size_t)l * Nh + h];
        }
    }
    part[ty][threadIdx.x] = make_double2(a, b);
    parti[ty][threadIdx.x] = bi;
    __syncthreads();
    if (ty != 0) return;
    bi = (parti[0][threadIdx.x] + parti[1][threadIdx.x]) + (parti[2][threadIdx.x] + parti[3][threadIdx.x]);
    a = (part[0][threadIdx.x].x + part[1][threadIdx.x].x) + (part[2][threadIdx.x].x + part[3][threadIdx.x].x);
    b = (part[0][threadIdx.x].y + part[1][threadIdx.x].y) + (part[2][threadIdx.x].y + part[3][threadIdx.x].y);
    if (j < N) {
        dbar[(size_t)w * N + j] = a / Lt;
    } else if (j < N + ptotal) {
        if (h >= 0) {
            a /= Lt; b /= Lt;
            cbar[(size_t)w * Nh + h] = a;
            sbar[(size_t)w * Nh + h] = b;
            if (sbari) sbari[(size_t)w * Nh + h] = bi / Lt;
        } else {
            a = 1.0; b = 0.0;  // identity self bond
        }
        pcs[(size_t)w * ptotal + (j - N)] = make_double2(a, b);
        if (pcsi) pcsi[(size_t)w * ptotal + (j - N)] = h >= 0 ? bi / Lt : 0.0;
    }
}

void launch_tau_means(hipStream_t st, const KpmGeom &kg, const double *expV, const double *ch, const double *sh, double *dbar, double *cbar, double *sbar, int Lt, int N, int Nh, int w0, int nw, const double *shi,
                      double *sbari)
{
    dim3 grid((N + kg.ptotal + 63) / 64, nw);
    hipLaunchKernelGGL(tau_means_kernel, grid, dim3(64, 4), 0, st, expV + (size_t)w0 * Lt * N, ch + (size_t)w0 * Lt * Nh, sh + (size_t)w0 * Lt * Nh, dbar + (size_t)w0 * N, cbar + (size_t)w0 * Nh,
                       sbar + (size_t)w0 * Nh, kg.pcs + (size_t)w0 * kg.ptotal, kg.psrc, kg.ptotal, Lt, N, Nh, shi ? shi + (size_t)w0 * Lt * Nh : nullptr, sbari ? sbari + (size_t)w0 * Nh : nullptr,
                       (kg.pcsi && shi) ? kg.pcsi + (size_t)w0 * kg.ptotal : nullptr);
}

// ---------------------------------------------------------------------------------------------
// register-resident bond program of one lane
// ---------------------------------------------------------------------------------------------
struct LaneBonds {
    int2 b[kMaxColours];  // LDS positions
    int2 s0, sL;          // site ids of the first / last colour's bond
    double2 cs[kMaxColours];
    double si[kMaxColours];  // Im s̄ for T = ComplexF64 (KpmGeom::pcsi), else 0 and unused
    bool on[kMaxColours];
};

__device__ __forceinline__ void load_lane_bonds(LaneBonds &lb, const KpmGeom &kg, int w, int ncol)
{
    lb.s0 = lb.sL = make_int2(0, 0);
    if (ncol >= 1 && kg.poff[0] + (int)threadIdx.x < kg.poff[1]) lb.s0 = kg.psites[kg.poff[0] + (int)threadIdx.x];
    if (ncol >= 1 && kg.poff[ncol - 1] + (int)threadIdx.x < kg.poff[ncol]) lb.sL = kg.psites[kg.poff[ncol - 1] + (int)threadIdx.x];
#pragma unroll
    for (int c = 0; c < kMaxColours; ++c) {
        lb.on[c] = false;
        lb.b[c] = make_int2(0, 0);
        lb.cs[c] = make_double2(1.0, 0.0);
        lb.si[c] = 0.0;
        if (c < ncol) {
            const int idx = kg.poff[c] + (int)threadIdx.x;
            if (idx < kg.poff[c + 1]) {
                lb.on[c] = true;
                lb.b[c] = kg.pbonds[idx];
                lb.cs[c] = kg.pcs[(size_t)w * kg.ptotal + idx];
                if (kg.pcsi) lb.si[c] = kg.pcsi[(size_t)w * kg.ptotal + idx];
            }
        }
    }
}

// plain colour stage on the LDS vector W
#define PLAIN_STAGE(c_)                                                         \
    {                                                                           \
        if (lb.on[c_]) {                                                        \
            const double2 a_ = W[lb.b[c_].x], d_ = W[lb.b[c_].y];               \
            W[lb.b[c_].x] = lin2(lb.cs[c_].x, a_, lb.cs[c_].y, d_);             \
            W[lb.b[c_].y] = lin2(lb.cs[c_].x, d_, lb.cs[c_].y, a_);             \
        }                                                                       \
        __syncthreads();                                                        \
    }

// the bond factor [[c, s], [conj(s), c]] on (a, d) = (u_i, u_j), bond (i, j) in neighbour-table order (checkerboard_matrix_multiply.jl:60-68);
// CPLX = false: s real, the plain pair of lin2
template <bool CPLX>
__device__ __forceinline__ void bond2(double c, double sr, double si, double2 a, double2 d, double2 &oa, double2 &od)
{
    oa = lin2(c, a, sr, d);
    od = lin2(c, d, sr, a);
    if (CPLX) {
        oa.x -= si * d.y; oa.y += si * d.x;
        od.x += si * a.y; od.y -= si * a.x;
    }
}
// PLAIN_STAGE with the bond factor of a complex s̄ when CPLX (a template parameter in scope) says so
#define PLAIN_STAGE_C(c_)                                                                         \
    {                                                                                             \
        if (lb.on[c_]) {                                                                          \
            const double2 a_ = W[lb.b[c_].x], d_ = W[lb.b[c_].y];                                 \
            double2 oa_, od_;                                                                     \
            bond2<CPLX>(lb.cs[c_].x, lb.cs[c_].y, lb.si[c_], a_, d_, oa_, od_);                   \
            W[lb.b[c_].x] = oa_;                                                                  \
            W[lb.b[c_].y] = od_;                                                                  \
        }                                                                                         \
        __syncthreads();                                                                          \
    }

// W <- B W where B = Sym B̄ (MODE 0), Asym B̄ = D̄Γ̄ (MODE 1) or Asym B̄ᵀB̄ = Γ̄ᵀD̄²Γ̄ (MODE 2); plain form
// used by Lanczos (no basis change)
template <int MODE>
__device__ __forceinline__ void bbar_apply_regs(double2 *W, const LaneBonds &lb, int ncol, int N, const double *__restrict__ dbar, const int *__restrict__ pos)
{
    if (MODE == 0) {
#pragma unroll
        for (int c = kMaxColours - 1; c >= 1; --c)
            if (c < ncol) PLAIN_STAGE(c)
        // C₁ D̄ C₁ fused (colour 0 is padded to cover every site)
        if (lb.on[0]) {
            const double2 a = W[lb.b[0].x], d = W[lb.b[0].y];
            const double di = dbar[lb.s0.x], dj = dbar[lb.s0.y];
            double2 x = lin2(lb.cs[0].x, a, lb.cs[0].y, d), y = lin2(lb.cs[0].x, d, lb.cs[0].y, a);
            x = make_double2(di * x.x, di * x.y);
            y = make_double2(dj * y.x, dj * y.y);
            W[lb.b[0].x] = lin2(lb.cs[0].x, x, lb.cs[0].y, y);
            W[lb.b[0].y] = lin2(lb.cs[0].x, y, lb.cs[0].y, x);
        }
        __syncthreads();
#pragma unroll
        for (int c = 1; c < kMaxColours; ++c)
            if (c < ncol) PLAIN_STAGE(c)
    } else {
#pragma unroll
        for (int c = 0; c < kMaxColours; ++c)
            if (c < ncol) PLAIN_STAGE(c)
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double d = (MODE == 2) ? dbar[i] * dbar[i] : dbar[i];
            const int q = pos[i];
            W[q] = make_double2(d * W[q].x, d * W[q].y);
        }
        __syncthreads();
        if (MODE == 2) {
#pragma unroll
            for (int c = kMaxColours - 1; c >= 0; --c)
                if (c < ncol) PLAIN_STAGE(c)
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fast Chebyshev kernel
// ---------------------------------------------------------------------------------------------
// One polynomial: on entry the lane holds the input at its last-colour site pair in (vi, vj);
// on exit (vi, vj) hold Σ_k coefs[k] T_k(B') v at the same sites.  W is the N-vector in LDS.
template <bool SYM, bool CPLX = false>
__device__ __forceinline__ void kpm_poly_regs(double2 *W, const LaneBonds &lb, int ncol, const double2 *__restrict__ coefs, int n, double avg, double imag_, double di, double dj, double dLi, double dLj,
                                              double2 &vi, double2 &vj)
{
    const int L = ncol;
    const int cl = L - 1;          // last colour (owner of the recurrence state)
    // last-colour data (runtime colour index -> select through an unrolled scan)
    int2 bL = make_int2(0, 0);
    double2 csL = make_double2(1.0, 0.0);
    double siL = 0.0;
    bool onL = false;
#pragma unroll
    for (int c = 0; c < kMaxColours; ++c)
        if (c == cl) { bL = lb.b[c]; csL = lb.cs[c]; siL = lb.si[c]; onL = lb.on[c]; }
    // Sym with L >= 2 runs in the basis α̃ = C_L α
    const bool xf = SYM && L >= 2;
    double2 a1i = vi, a1j = vj;
    if (xf) bond2<CPLX>(csL.x, csL.y, siL, vi, vj, a1i, a1j);
    if (onL) { W[bL.x] = a1i; W[bL.y] = a1j; }
    __syncthreads();
    // C_L² = [[c² + |s|², 2cs], [2c conj(s), c² + |s|²]]
    const double q_c = CPLX ? csL.x * csL.x + csL.y * csL.y + siL * siL : csL.x * csL.x + csL.y * csL.y, q_s = 2.0 * csL.x * csL.y, q_si = CPLX ? 2.0 * csL.x * siL : 0.0;
    double2 a2i = make_double2(0, 0), a2j = a2i, acci = a2i, accj = a2i;
    for (int k = 1; k < n; ++k) {
        const double2 ck = coefs[k];  // issued early; consumed after the stages
        double2 xi, xj;
        if (SYM) {
#pragma unroll
            for (int c = kMaxColours - 2; c >= 1; --c)
                if (c < L - 1) PLAIN_STAGE_C(c)
            if (L >= 2) {
                if (lb.on[0]) {
                    const double2 a = W[lb.b[0].x], d = W[lb.b[0].y];
                    double2 x, y, ox, oy;
                    bond2<CPLX>(lb.cs[0].x, lb.cs[0].y, lb.si[0], a, d, x, y);
                    x = make_double2(di * x.x, di * x.y);
                    y = make_double2(dj * y.x, dj * y.y);
                    bond2<CPLX>(lb.cs[0].x, lb.cs[0].y, lb.si[0], x, y, ox, oy);
                    W[lb.b[0].x] = ox;
                    W[lb.b[0].y] = oy;
                }
                __syncthreads();
#pragma unroll
                for (int c = 1; c < kMaxColours - 1; ++c)
                    if (c < L - 1) PLAIN_STAGE_C(c)
                const double2 a = W[bL.x], d = W[bL.y];
                bond2<CPLX>(q_c, q_s, q_si, a, d, xi, xj);
            } else {  // single colour: B̄ = C₁ D̄ C₁, state owned by colour 0 itself
                const double2 a = W[bL.x], d = W[bL.y];
                double2 x, y;
                bond2<CPLX>(csL.x, csL.y, siL, a, d, x, y);
                x = make_double2(di * x.x, di * x.y);
                y = make_double2(dj * y.x, dj * y.y);
                bond2<CPLX>(csL.x, csL.y, siL, x, y, xi, xj);
            }
        } else {
#pragma unroll
            for (int c = 0; c < kMaxColours - 1; ++c)
                if (c < L - 1) PLAIN_STAGE_C(c)
            const double2 a = W[bL.x], d = W[bL.y];
            bond2<CPLX>(csL.x, csL.y, siL, a, d, xi, xj);
            xi = make_double2(dLi * xi.x, dLi * xi.y);
            xj = make_double2(dLj * xj.x, dLj * xj.y);
        }
        // three-term recurrence on the lane's own site pair (kpm_lmul!)
        double2 a3i, a3j;
        if (k == 1) {
            a3i = make_double2((xi.x - avg * a1i.x) * imag_, (xi.y - avg * a1i.y) * imag_);
            a3j = make_double2((xj.x - avg * a1j.x) * imag_, (xj.y - avg * a1j.y) * imag_);
            const double2 c0 = coefs[0];
            const double2 t0i = cmulk(c0, a1i), t0j = cmulk(c0, a1j), t1i = cmulk(ck, a3i), t1j = cmulk(ck, a3j);
            acci = make_double2(t0i.x + t1i.x, t0i.y + t1i.y);
            accj = make_double2(t0j.x + t1j.x, t0j.y + t1j.y);
            a2i = a3i; a2j = a3j;
        } else {
            a3i = make_double2(2.0 * (xi.x - avg * a2i.x) * imag_ - a1i.x, 2.0 * (xi.y - avg * a2i.y) * imag_ - a1i.y);
            a3j = make_double2(2.0 * (xj.x - avg * a2j.x) * imag_ - a1j.x, 2.0 * (xj.y - avg * a2j.y) * imag_ - a1j.y);
            const double2 ti = cmulk(ck, a3i), tj = cmulk(ck, a3j);
            acci = make_double2(acci.x + ti.x, acci.y + ti.y);
            accj = make_double2(accj.x + tj.x, accj.y + tj.y);
            a1i = a2i; a1j = a2j;
            a2i = a3i; a2j = a3j;
        }
        if (k + 1 < n) {
            if (onL) { W[bL.x] = a3i; W[bL.y] = a3j; }
            __syncthreads();
        }
    }
    if (xf) {  // back to the original basis: C_L⁻¹
        if (CPLX) {  // C_L⁻¹ = [[c, −s], [−conj(s), c]] / (c² − |s|²)
            const double idet = 1.0 / (csL.x * csL.x - csL.y * csL.y - siL * siL);
            bond2<true>(csL.x, -csL.y, -siL, acci, accj, vi, vj);
            vi = make_double2(vi.x * idet, vi.y * idet);
            vj = make_double2(vj.x * idet, vj.y * idet);
        } else {
            const double idet = 1.0 / (csL.x * csL.x - csL.y * csL.y);
            vi = make_double2((csL.x * acci.x - csL.y * accj.x) * idet, (csL.x * acci.y - csL.y * accj.y) * idet);
            vj = make_double2((csL.x * accj.x - csL.y * acci.x) * idet, (csL.x * accj.y - csL.y * acci.y) * idet);
        }
    } else {
        vi = acci;
        vj = accj;
    }
}

// workgroup sum of a complex value, broadcast (red: >= 17 doubles of LDS)
__device__ __forceinline__ double2 block_sum_cplx(double2 v, double *red)
{
    const double re = block_sum_real(v.x, red);
    const double im = block_sum_real(v.y, red);
    return make_double2(re, im);
}

// NCOL > 0 fixes the number of colours at compile time (the per-stage colour tests fold away);
// NCOL = 0 reads it at run time.
template <bool SYM, int NCOL, bool CPLX = false>
__global__ void __launch_bounds__(1024) cheb_fast_kernel(KpmArgs k, KpmGeom kg)
{
    const int ncol = NCOL > 0 ? NCOL : k.ncol;
    extern __shared__ double2 lds[];
    __shared__ double red[17];
    double2 *W = lds;
    const int N = k.N, Lt = k.Lt;
    // rank-major block order: the heaviest chains of all systems are dispatched first and land
    // on different CUs / XCDs (a system-major order parks them on the same CU)
    const int ncnt_ = k.sys_count > 0 ? k.sys_count : k.nsys;
    const int sys = k.sys_first + blockIdx.x % ncnt_, rank = blockIdx.x / ncnt_;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);  // heaviest orders first
    const int w = sys / k.nrhs;
    if (k.cg[sys].done) return;  // k.cg is never null (api_handle.hip, fdm_args / kpm_args: an all-zero state outside CG loops)
    const double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)om * k.nsys + sys) * N;
    double2 *prz = k.part_rz ? k.part_rz + (size_t)sys * k.rz_stride + om : nullptr;
    const int Lo2 = (Lt + 1) / 2;
    if (k.half && om >= Lo2) return;
    const int slot = SYM ? (om >= Lo2 ? Lt - om - 1 : om) : om;  // :387
    const bool act = k.active[w] != 0;                           // inactive preconditioner = identity (:410)
    const int n = act ? k.order[(size_t)w * k.nslot + slot] : 1;
    const double2 *coefs = k.coefs + ((size_t)w * k.nslot + slot) * k.maxorder;
    if (n <= 1) {
        // single-term expansion: scalar multiply (:398 / :534)
        double f = k.scale;
        if (act) {
            const double2 c0 = coefs[0];
            f *= SYM ? c0.x : (c0.x * c0.x + c0.y * c0.y);
        }
        double acc = 0.0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double2 x = v[i];
            vo[i] = make_double2(f * x.x, f * x.y);
            acc += f * (x.x * x.x + x.y * x.y);
        }
        if (prz) {
            const double t = block_sum_real(acc, red);
            if (threadIdx.x == 0) *prz = make_double2(t, 0.0);
        }
        return;
    }
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    const double avg = 0.5 * (emax + emin), imag_ = 1.0 / (0.5 * (emax - emin));
    LaneBonds lb;
    load_lane_bonds(lb, kg, w, ncol);
    const double *dbar = k.dbar + (size_t)w * N;
    const int cl = ncol - 1;
    int2 bL = make_int2(0, 0);
    bool onL = false;
#pragma unroll
    for (int c = 0; c < kMaxColours; ++c)
        if (c == cl) { bL = lb.b[c]; onL = lb.on[c]; }
    const double di = lb.on[0] ? dbar[lb.s0.x] : 1.0, dj = lb.on[0] ? dbar[lb.s0.y] : 1.0;
    const double dLi = onL ? dbar[lb.sL.x] : 1.0, dLj = onL ? dbar[lb.sL.y] : 1.0;
    double2 vi = make_double2(0, 0), vj = vi;
    if (onL) { vi = v[lb.sL.x]; vj = v[lb.sL.y]; }
    const double2 vi0 = vi, vj0 = vj;
    // expansion coefficients go to LDS once: no global load (and no vmcnt wait) inside the chain
    double2 *CF = W + N, *CF2 = CF + k.maxorder;
    const int omc = Lt - om - 1;  // :523-530
    const int n2 = SYM ? 0 : k.order[(size_t)w * k.nslot + omc];
    for (int i = threadIdx.x; i < n; i += blockDim.x) CF[i] = coefs[i];
    if (!SYM) {
        const double2 *coefs_c = k.coefs + ((size_t)w * k.nslot + omc) * k.maxorder;
        for (int i = threadIdx.x; i < n2; i += blockDim.x) CF2[i] = coefs_c[i];
    }
    __syncthreads();
    if (SYM) {
        kpm_poly_regs<true, CPLX>(W, lb, ncol, CF, n, avg, imag_, di, dj, dLi, dLj, vi, vj);  // :394
    } else {
        kpm_poly_regs<false, CPLX>(W, lb, ncol, CF2, n2, avg, imag_, di, dj, dLi, dLj, vi, vj);
        __syncthreads();
        kpm_poly_regs<false, CPLX>(W, lb, ncol, CF, n, avg, imag_, di, dj, dLi, dLj, vi, vj);
    }
    double2 acc = make_double2(0.0, 0.0);
    if (onL) {
        vi = make_double2(k.scale * vi.x, k.scale * vi.y);
        vj = make_double2(k.scale * vj.x, k.scale * vj.y);
        vo[lb.sL.x] = vi;
        acc.x += vi0.x * vi.x + vi0.y * vi.y;
        acc.y += vi0.x * vi.y - vi0.y * vi.x;
        if (bL.y != bL.x) {
            vo[lb.sL.y] = vj;
            acc.x += vj0.x * vj.x + vj0.y * vj.y;
            acc.y += vj0.x * vj.y - vj0.y * vj.x;
        }
    }
    if (prz) {
        const double2 t = block_sum_cplx(acc, red);
        if (threadIdx.x == 0) *prz = t;
    }
}


// ---------------------------------------------------------------------------------------------
// owner-computes Chebyshev kernel (Sym, 2..kMaxColours colours)
// ---------------------------------------------------------------------------------------------
// cheb_fast_kernel pays one LDS round trip + barrier per colour stage (2L-2 per Chebyshev step) because a stage
// reads its pair, updates it and writes it back for the next colour's lanes.  Here lane j OWNS the two sites of
// the j-th padded bond of colour q for the whole chain and updates both of them itself in every stage: a stage
// of colour q is pure register arithmetic, and a stage of another colour needs one exchange (own values to
// LDS, barrier, read the two mates) followed by  a' = c·a + s·mate(a)  for each own site with that site's bond.
// q is a colour that occurs twice per step (colour 1 for L >= 3; colour 0 for L = 2), which leaves 2L-4 (L >= 3)
// or 1 (L = 2) exchanges per step — two instead of four on the honeycomb lattice.  The fused C₁D̄C₁ stage
// recomputes the mate's intermediate value (same bond, mate's d̄) instead of fetching it.  Exchanges ping-pong
// between two LDS images, so each costs a single barrier.  Arithmetic per site is identical to cheb_fast_kernel.
// SPLIT = true ("component split"): B̄, the bounds and the Sym coefficients are all REAL, so the real and the imaginary part of a frequency
// vector are two independent real recurrences.  Each gets its own workgroup: a lane then carries half the arithmetic per exchange
// (a step costs 0.58 synchronisation + 0.42 instruction issue, measured through the paired-frequency experiment), the
// LDS images shrink to doubles, and the longest chain — which IS the kernel's duration — gets shorter.  Per component the arithmetic
// is exactly that of the complex kernel (products with an exact zero imaginary part drop out), so the output is bit-identical; only
// the Parseval partial of r·z is accumulated in two slots per frequency instead of one, and its imaginary part (rounding noise around
// the exact zero of a real symmetric P⁻¹) is not formed.
namespace ownk {
template <bool SPLIT> struct Scalar { using type = double2; };
template <> struct Scalar<true> { using type = double; };
__device__ __forceinline__ double2 zero(double2) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ double zero(double) { return 0.0; }
__device__ __forceinline__ double lin(double a, double x, double b, double y) { return a * x + b * y; }
__device__ __forceinline__ double2 lin(double a, double2 x, double b, double2 y) { return lin2(a, x, b, y); }
__device__ __forceinline__ double scl(double a, double x) { return a * x; }
__device__ __forceinline__ double2 scl(double a, double2 x) { return make_double2(a * x.x, a * x.y); }
__device__ __forceinline__ double add(double x, double y) { return x + y; }
__device__ __forceinline__ double2 add(double2 x, double2 y) { return make_double2(x.x + y.x, x.y + y.y); }
__device__ __forceinline__ double sub(double x, double y) { return x - y; }
__device__ __forceinline__ double2 sub(double2 x, double2 y) { return make_double2(x.x - y.x, x.y - y.y); }
// coefficient times vector: the complex kernel keeps the general complex product (Sym coefficients have an exact zero imaginary part)
__device__ __forceinline__ double cmul(double2 c, double x) { return c.x * x; }
__device__ __forceinline__ double2 cmul(double2 c, double2 x) { return cmulk(c, x); }
__device__ __forceinline__ double shfl(double x, int lane) { return __shfl(x, lane, 64); }
__device__ __forceinline__ double2 shfl(double2 x, int lane) { return make_double2(__shfl(x.x, lane, 64), __shfl(x.y, lane, 64)); }
// rotation within rows of 16 lanes as a DPP modifier (a register move — no trip through the LDS permute unit as for ds_bpermute):
// CTRL = 0x120 + n is row_ror:n, lane i takes the value of lane (i - n) mod 16 of its row
template <int CTRL>
__device__ __forceinline__ double row_rot(double x)
{
    // a rotation gives every lane a source lane: mov_dpp, no "old" register to clear first
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double2 row_rot(double2 x) { return make_double2(row_rot<CTRL>(x.x), row_rot<CTRL>(x.y)); }
// the two colour-0 mates of a lane under KpmGeom::wl0 = WL0: 1 — any lanes of the wavefront (ds_bpermute); 2 / 3 — the neighbours in the
// lane's row of 16, cyclically (mate of the first site one lane down and of the second one lane up, or the other way round)
template <int WL0, class T>
__device__ __forceinline__ void wl_mates(T ax, T ay, int wl_x, int wl_y, T &mx, T &my)
{
    if constexpr (WL0 == 2) { mx = row_rot<0x121>(ay); my = row_rot<0x12F>(ax); }
    else if constexpr (WL0 == 3) { mx = row_rot<0x12F>(ay); my = row_rot<0x121>(ax); }
    else { mx = shfl(ay, wl_x); my = shfl(ax, wl_y); }
}
__device__ __forceinline__ double ld(const double2 *v, int i, int comp, double) { return comp ? v[i].y : v[i].x; }
__device__ __forceinline__ double2 ld(const double2 *v, int i, int, double2) { return v[i]; }
__device__ __forceinline__ void st(double2 *v, int i, int comp, double x) { if (comp) v[i].y = x; else v[i].x = x; }
__device__ __forceinline__ void st(double2 *v, int i, int, double2 x) { v[i] = x; }
}  // namespace ownk

// the lane program of the owner-computes kernels: the lane's two sites, their LDS slots, per colour the mates' slots and the (c̄, s̄) of the two
// bonds, the τ-means of exp(-ΔτV) at the own sites and at their colour-0 mates.  It depends on the walker only.
template <int NCOL>
struct OwnProg {
    bool on;
    int sx, sy, ox, oy;
    int px[NCOL], py[NCOL];
    double2 cx[NCOL], cy[NCOL];
    double dx, dy, dmx, dmy;
    // the fused centre stage C₀ D̄ C₀ as two coefficients per own site (round 4: 4 instead of 16 fp64 operations per lane and stage).
    // Colour 0 foreign (NCOL >= 3):  a' = (c² d + s² d_mate) a + c s (d + d_mate) mate   with the site's colour-0 bond (c, s);
    // colour 0 owned (NCOL = 2):     ax' = ex0 ax + ex1 ay,  ay' = ex1 ax + ey0 ay       (the 2x2 matrix C D̄ C of the own bond).
    double ex0, ex1, ey0, ey1;
};
// table indices of a lane program: the first of its two rounds of loads (addresses known from the thread index alone)
template <int NCOL>
struct OwnIdx {
    bool on;
    int sx, sy, m2, m3;
    int px[NCOL], py[NCOL], cxi[NCOL], cyi[NCOL];
};
template <int NCOL>
__device__ __forceinline__ void own_load_idx(OwnIdx<NCOL> &I, const KpmGeom &kg, int Tn, int j)
{
    I.on = j < kg.own_n;
    const int *own = kg.own;
    const int jq = I.on ? j : 0;  // lanes past the list load entry 0 and ignore it
    I.sx = own[jq]; I.sy = own[Tn + jq]; I.m2 = own[2 * Tn + jq]; I.m3 = own[3 * Tn + jq];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        I.px[c] = own[(4 + 4 * c + 0) * Tn + jq];
        I.py[c] = own[(4 + 4 * c + 1) * Tn + jq];
        I.cxi[c] = own[(4 + 4 * c + 2) * Tn + jq];
        I.cyi[c] = own[(4 + 4 * c + 3) * Tn + jq];
    }
}
// second round: the gathers addressed through the first
template <int NCOL>
__device__ __forceinline__ void own_load_prog(OwnProg<NCOL> &P, const OwnIdx<NCOL> &I, const double *__restrict__ dbar, const double2 *__restrict__ pcs, int Tn, int j)
{
    P.on = I.on;
    P.sx = P.sy = 0; P.ox = P.oy = j;
    P.dx = P.dy = P.dmx = P.dmy = 1.0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) { P.px[c] = P.py[c] = j; P.cx[c] = P.cy[c] = make_double2(1.0, 0.0); }
    if (I.on) {
        P.sx = I.sx; P.sy = I.sy;
        P.oy = (I.sy != I.sx) ? Tn + j : j;
        P.dx = dbar[I.sx]; P.dy = dbar[I.sy];
        P.dmx = dbar[I.m2]; P.dmy = dbar[I.m3];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            P.px[c] = I.px[c];
            P.py[c] = I.py[c];
            P.cx[c] = pcs[I.cxi[c]];
            P.cy[c] = pcs[I.cyi[c]];
        }
    }
    if constexpr (NCOL >= 3) {
        P.ex0 = P.cx[0].x * P.cx[0].x * P.dx + P.cx[0].y * P.cx[0].y * P.dmx;
        P.ex1 = P.cx[0].x * P.cx[0].y * (P.dx + P.dmx);
        P.ey0 = P.cy[0].x * P.cy[0].x * P.dy + P.cy[0].y * P.cy[0].y * P.dmy;
        P.ey1 = P.cy[0].x * P.cy[0].y * (P.dy + P.dmy);
    } else {
        P.ex0 = P.cx[0].x * P.cx[0].x * P.dx + P.cx[0].y * P.cx[0].y * P.dy;
        P.ey0 = P.cx[0].y * P.cx[0].y * P.dx + P.cx[0].x * P.cx[0].x * P.dy;
        P.ex1 = P.ey1 = P.cx[0].x * P.cx[0].y * (P.dx + P.dy);
    }
}

// Exchange and stage of the owner-computes lane program (used inside functions that define P, ax, ay, Wb0, Wb1, buf and the constants
// Q = owned colour, T = value type): own values -> LDS image, ONE barrier (the images ping-pong), the two mates of colour c_ come back.
#define OWN_EXCHANGE(c_, mx_, my_)                       \
    {                                                    \
        T *Wc = buf ? Wb1 : Wb0;                         \
        buf ^= 1;                                        \
        if (P.on) { Wc[P.ox] = ax; Wc[P.oy] = ay; }      \
        __syncthreads();                                 \
        mx_ = Wc[P.px[c_]];                              \
        my_ = Wc[P.py[c_]];                              \
    }
#define OWN_STAGE(c_)                                                        \
    {                                                                        \
        if (c_ == Q) {                                                       \
            const T t_ = lin(P.cx[Q].x, ax, P.cx[Q].y, ay);                  \
            ay = lin(P.cx[Q].x, ay, P.cx[Q].y, ax);                          \
            ax = t_;                                                         \
        } else {                                                             \
            T mx_, my_;                                                      \
            OWN_EXCHANGE(c_, mx_, my_)                                       \
            ax = lin(P.cx[c_].x, ax, P.cx[c_].y, mx_);                       \
            ay = lin(P.cy[c_].x, ay, P.cy[c_].y, my_);                       \
        }                                                                    \
    }
// the fused centre stage C₀ D̄ C₀ on the own sites: register arithmetic when colour 0 is the owned one, otherwise one exchange with the
// mate's intermediate value recomputed here (same bond, the mate's own d̄)
#define OWN_CENTRE()                                                                                          \
    {                                                                                                         \
        if (Q == 0) {                                                                                         \
            const T x_ = lin(P.ex0, ax, P.ex1, ay);                                                           \
            ay = lin(P.ey1, ax, P.ey0, ay);                                                                   \
            ax = x_;                                                                                          \
        } else {                                                                                              \
            T mx_, my_;                                                                                       \
            OWN_EXCHANGE(0, mx_, my_)                                                                         \
            ax = lin(P.ex0, ax, P.ex1, mx_);                                                                  \
            ay = lin(P.ey0, ay, P.ey1, my_);                                                                  \
        }                                                                                                     \
    }

// (ax, ay) <- B̄ (ax, ay) for the Sym propagator in its plain form B̄ = C_{L-1} … C_1 (C_0 D̄ C_0) C_1 … C_{L-1} (no basis change: what
// Lanczos needs, KPMPreconditioner.jl:625-639) on the lane's own two sites; all lanes of the workgroup call it together
template <int NCOL, class T, int WL0 = 0>
__device__ __forceinline__ void own_bbar_apply(const OwnProg<NCOL> &P, T &ax, T &ay, T *Wb0, T *Wb1, int &buf)
{
    using namespace ownk;
    constexpr int Q = NCOL >= 3 ? 1 : 0;
#pragma unroll
    for (int c = NCOL - 1; c >= 1; --c) OWN_STAGE(c)
    if constexpr (WL0 && Q != 0) {  // the colour-0 mates sit in this wavefront (KpmGeom::wl0): shuffles instead of an LDS exchange
        T mx_, my_;
        wl_mates<WL0>(ax, ay, (P.px[0] - (int)blockDim.x) & 63, P.py[0] & 63, mx_, my_);
        ax = lin(P.ex0, ax, P.ex1, mx_);
        ay = lin(P.ey0, ay, P.ey1, my_);
    } else
    OWN_CENTRE()
#pragma unroll
    for (int c = 1; c <= NCOL - 1; ++c) OWN_STAGE(c)
}

// Σ_k CF[k] T_k(B̄') applied to the values (ax, ay) of the lane's own sites, n >= 2 terms (kpm_lmul!, Sym B̄ in the basis α̃ = C_L α).
// T = double: one component of the vector (SPLIT), T = double2: both; Wb are two LDS images of 2·Tn values of T; all lanes of the
// workgroup call it together (it contains barriers); CF must be visible to all lanes on entry.
// WL0 (round 3): the host has verified that the colour-0 mates of every lane's two sites sit in lanes of the SAME wavefront, the mate of
// the first site in some lane's second slot and vice versa (KpmGeom::wl0; honeycomb L = 16: the colour-0 partner is the neighbouring cell
// of the same row of 16).  The centre exchange of a Chebyshev step then needs no LDS image and no barrier: two wave shuffles
// (ds_bpermute) bring the mates.  Same values, same arithmetic: bit-identical to the LDS form.
template <int NCOL, class T, int WL0 = 0>
__device__ __forceinline__ void own_chain(const OwnProg<NCOL> &P, T &ax, T &ay, T *Wb0, T *Wb1, int &buf, const double2 *CF, int n, double avg, double imag_)
{
    using namespace ownk;
    constexpr int Q = NCOL >= 3 ? 1 : 0, CL = NCOL - 1;
    const int wl_x = (P.px[0] - (int)blockDim.x) & 63, wl_y = P.py[0] & 63;  // source lanes of the two colour-0 mates (used with WL0 only)
    (void)wl_x; (void)wl_y;
    // into the basis α̃ = C_L α (see cheb_fast_kernel)
    {
        T mx, my;
        OWN_EXCHANGE(CL, mx, my)
        ax = lin(P.cx[CL].x, ax, P.cx[CL].y, mx);
        ay = lin(P.cy[CL].x, ay, P.cy[CL].y, my);
    }
    const double qcx = P.cx[CL].x * P.cx[CL].x + P.cx[CL].y * P.cx[CL].y, qsx = 2.0 * P.cx[CL].x * P.cx[CL].y;  // C_L²
    const double qcy = P.cy[CL].x * P.cy[CL].x + P.cy[CL].y * P.cy[CL].y, qsy = 2.0 * P.cy[CL].x * P.cy[CL].y;
    const double imag2 = 2.0 * imag_;
    // (xi, xj) = B̃ (ax, ay) on the lane's own sites; ax, ay are used up
    auto apply = [&](T ax, T ay, T &xi, T &xj) {
#pragma unroll
        for (int c = NCOL - 2; c >= 1; --c) OWN_STAGE(c)
        if (Q == 0) {  // C₀ D̄ C₀ in registers: the 2x2 matrix of the own bond
            const T x = lin(P.ex0, ax, P.ex1, ay);
            ay = lin(P.ey1, ax, P.ey0, ay);
            ax = x;
        } else {       // one exchange; the mate's part of C₀ D̄ C₀ is folded into the two coefficients of the site (OwnProg::ex0 …)
            T mx, my;
            if constexpr (WL0) {
                wl_mates<WL0>(ax, ay, wl_x, wl_y, mx, my);
            } else {
                OWN_EXCHANGE(0, mx, my)
            }
            ax = lin(P.ex0, ax, P.ex1, mx);
            ay = lin(P.ey0, ay, P.ey1, my);
        }
#pragma unroll
        for (int c = 1; c <= NCOL - 2; ++c) OWN_STAGE(c)
        T mx, my;
        OWN_EXCHANGE(CL, mx, my)
        xi = lin(qcx, ax, qsx, mx);
        xj = lin(qcy, ay, qsy, my);
    };
    // three-term recurrence on the lane's own sites (kpm_lmul!): T_k = 2 B' T_{k-1} − T_{k-2} is written over T_{k-2}, so the two newest
    // vectors ping-pong between (ax, ay) and (bx, by) and no register is moved (round 4)
    T bx, by, accx, accy, xi, xj;
    {   // k = 1
        apply(ax, ay, xi, xj);
        const double2 c0 = CF[0], c1 = CF[1];
        bx = scl(imag_, sub(xi, scl(avg, ax)));
        by = scl(imag_, sub(xj, scl(avg, ay)));
        accx = add(cmul(c0, ax), cmul(c1, bx));
        accy = add(cmul(c0, ay), cmul(c1, by));
    }
#define OWN_STEP(cx_, cy_, ox_, oy_, ck_)                              \
    {                                                                  \
        apply(cx_, cy_, xi, xj);                                       \
        ox_ = sub(scl(imag2, sub(xi, scl(avg, cx_))), ox_);            \
        oy_ = sub(scl(imag2, sub(xj, scl(avg, cy_))), oy_);            \
        accx = add(accx, cmul(ck_, ox_));                              \
        accy = add(accy, cmul(ck_, oy_));                              \
    }
    int kk = 2;
    for (; kk + 1 < n; kk += 2) {
        OWN_STEP(bx, by, ax, ay, CF[kk])
        OWN_STEP(ax, ay, bx, by, CF[kk + 1])
    }
    if (kk < n) OWN_STEP(bx, by, ax, ay, CF[kk])
#undef OWN_STEP
    // back to the original basis: C_L⁻¹ on the accumulated sum
    {
        ax = accx; ay = accy;
        T mx, my;
        OWN_EXCHANGE(CL, mx, my)
        const double idx_ = 1.0 / (P.cx[CL].x * P.cx[CL].x - P.cx[CL].y * P.cx[CL].y), idy_ = 1.0 / (P.cy[CL].x * P.cy[CL].x - P.cy[CL].y * P.cy[CL].y);
        ax = scl(idx_, sub(scl(P.cx[CL].x, accx), scl(P.cx[CL].y, mx)));
        ay = scl(idy_, sub(scl(P.cy[CL].x, accy), scl(P.cy[CL].y, my)));
    }
}

// Workgroups of a launch, per system: the k.heavy frequencies of lowest |ϕ| — every frequency whose expansion has more than one term; the
// order falls off like 1/ϕ (KPMPreconditioner.jl:709-711) and the host, which computed the orders, passes the count — get a workgroup each
// (two with SPLIT, one per component) and run their chain.  All the other frequencies are a scalar multiple of their vector (:398) and are
// handled k.group at a time by "light" workgroups, whole 16-byte elements, both components at once.  Done by a workgroup of its own a
// light frequency costs ≈ 2 µs of a workgroup slot for 16 KB of traffic — the round trips in front of the work — and with a workgroup per
// frequency and component the Chebyshev kernel held more wave-slot time than any other kernel of the iteration (SQ_WAVE_CYCLES 35.4 M
// against 31.9 M for MᵀM, profiles/r02_pmc_lds_iteration.txt), which is what the multi-stream bench is bound by.
template <int NCOL, bool SPLIT, int WL0 = 0>
__global__ void __launch_bounds__(1024) cheb_own_kernel(KpmArgs k, KpmGeom kg)
{
    static_assert(NCOL >= 2, "single-colour decompositions use cheb_fast_kernel");
    using namespace ownk;
    using T = typename Scalar<SPLIT>::type;
    extern __shared__ double2 lds[];
    __shared__ double red[17];
    const int N = k.N, Lt = k.Lt, Tn = blockDim.x, j = threadIdx.x;
    const int ncnt_ = k.sys_count > 0 ? k.sys_count : k.nsys;
    int sys = k.sys_first + blockIdx.x % ncnt_, slotid = blockIdx.x / ncnt_;
    if (k.xcd_map && (ncnt_ & 7) == 0) {
        // XCD x (workgroups go round-robin) takes the contiguous share [x·n/8, (x+1)·n/8) of the systems, as in the MᵀM and τ-FFT kernels;
        // within an XCD the order stays slot-major (heaviest chains first)
        const int per_ = ncnt_ >> 3, q_ = blockIdx.x >> 3;
        sys = k.sys_first + (blockIdx.x & 7) * per_ + q_ % per_;
        slotid = q_ / per_;
    }
    const int heavy = min(k.heavy, Lt), heavy_slots = SPLIT ? 2 * heavy : heavy;
    const int w = sys / k.nrhs;
    const int Lo2 = (Lt + 1) / 2;
    double2 *przb = k.part_rz ? k.part_rz + (size_t)sys * k.rz_stride : nullptr;
    if (slotid >= heavy_slots) {
        cheb_light_workgroup<SPLIT>(k, sys, w, slotid - heavy_slots, heavy, przb, lds);
        return;
    }
    // ---- heavy workgroup: one frequency (one component of it with SPLIT), heaviest first, the two components neighbours in the dispatch order ----
    T *W0 = reinterpret_cast<T *>(lds), *W1 = W0 + 2 * Tn;
    double2 *CF = reinterpret_cast<double2 *>(W0 + 4 * Tn);
    const int comp = SPLIT ? (slotid & 1) : 0, rank = SPLIT ? (slotid >> 1) : slotid;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);
    if (k.half && om >= Lo2) return;
    const int slot = om >= Lo2 ? Lt - om - 1 : om;  // :387
    // Round 1 of loads — everything addressed by the block and thread index alone, issued before any of it is looked at: the stop flag,
    // the expansion order, the spectral bounds and the lane program's table indices.  The workgroup with the longest chain sets the
    // duration of the launch and every serial round trip in front of the chain adds to it; as written naively (flag, return; active,
    // order, return; bounds; table; gathers) that was five.
    const bool sys_done = k.cg[sys].done != 0;  // (k.cg is never null)
    const bool act = k.active[w] != 0;
    const int n_raw = k.order[(size_t)w * k.nslot + slot];
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    OwnIdx<NCOL> I;
    own_load_idx<NCOL>(I, kg, Tn, j);
    asm volatile("" ::: "memory");  // compiler fence: keeps the loads above on this side of the early returns (no instruction, no wait)
    if (sys_done) return;
    const double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)om * k.nsys + sys) * N;
    double2 *prz = przb ? przb + (SPLIT ? 2 * om + comp : om) : nullptr;
    const int n = act ? n_raw : 1;
    const double2 *coefs = k.coefs + ((size_t)w * k.nslot + slot) * k.maxorder;
    if (n <= 1) {  // single-term expansion: scalar multiply (:398)
        const double f = k.scale * (act ? coefs[0].x : 1.0);
        double acc = 0.0;
        for (int i = j; i < N; i += Tn) {
            const T x = ld(v, i, comp, T{});
            st(vo, i, comp, scl(f, x));
            if constexpr (SPLIT) acc += f * (x * x);
            else acc += f * (x.x * x.x + x.y * x.y);
        }
        if (prz) {
            const double t = block_sum_real(acc, red);
            if (j == 0) *prz = make_double2(t, 0.0);
        }
        return;
    }
    const double avg = 0.5 * (emax + emin), imag_ = 1.0 / (0.5 * (emax - emin));
    // Round 2: the gathers addressed through round 1 (τ-means at the own sites and their mates, the bonds' (c̄, s̄), the input vector)
    OwnProg<NCOL> P;
    own_load_prog<NCOL>(P, I, k.dbar + (size_t)w * N, kg.pcs + (size_t)w * kg.ptotal, Tn, j);
    T ax = zero(T{}), ay = zero(T{});
    if (P.on) { ax = ld(v, P.sx, comp, T{}); ay = ld(v, P.sy, comp, T{}); }
    const T v0x = ax, v0y = ay;
    for (int i = j; i < n; i += Tn) CF[i] = coefs[i];  // coefficients in LDS: no global load inside the chain
    __syncthreads();  // CF visible
    int buf = 0;
    own_chain<NCOL, T, WL0>(P, ax, ay, W0, W1, buf, CF, n, avg, imag_);
    double2 acc = make_double2(0.0, 0.0);
    if (P.on) {
        ax = scl(k.scale, ax);
        ay = scl(k.scale, ay);
        st(vo, P.sx, comp, ax);
        if (P.sy != P.sx) st(vo, P.sy, comp, ay);
        if constexpr (SPLIT) {
            // Re conj(r)·z = r_re z_re + r_im z_im: this workgroup adds the term of its component.  The imaginary part
            // r_re z_im − r_im z_re couples the two workgroups of a frequency (and, in place, races with the other one's stores); for the
            // real symmetric P⁻¹ of the Sym form it is rounding noise around an exact zero and is left at zero.
            acc.x += v0x * ax;
            if (P.sy != P.sx) acc.x += v0y * ay;
        } else {
            acc.x += v0x.x * ax.x + v0x.y * ax.y;
            acc.y += v0x.x * ax.y - v0x.y * ax.x;
            if (P.sy != P.sx) {
                acc.x += v0y.x * ay.x + v0y.y * ay.y;
                acc.y += v0y.x * ay.y - v0y.y * ay.x;
            }
        }
    }
    if (prz) {
        if constexpr (SPLIT) {  // the imaginary part is not formed: one reduction pass
            const double t = block_sum_real(acc.x, red);
            if (j == 0) *prz = make_double2(t, 0.0);
        } else {
            const double2 t = block_sum_cplx(acc, red);
            if (j == 0) *prz = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// owner-computes Chebyshev kernel, Asym form (KPMPreconditioner.jl:514-536): per frequency two polynomials in B̄ = D̄ C_{L-1} … C_0, first with
// the coefficients of the mirrored frequency Lτ−1−ω, then with its own (M̃⁻¹ M̃⁻ᵀ).  Every colour occurs once per B̄ apply, so the owned colour
// (the same tables as the Sym kernel: colour 1 for L >= 3, colour 0 for L = 2) is register arithmetic and each of the other L−1 colours is
// one exchange — two per step on the honeycomb lattice where cheb_fast_kernel<Asym> pays three LDS read-modify-write stages with a
// barrier each.  The coefficients are complex, so the two components of a vector do not separate (no SPLIT form); arithmetic per site is
// that of kpm_poly_regs<false>.
// ---------------------------------------------------------------------------------------------
template <int NCOL>
__global__ void __launch_bounds__(1024) cheb_own_asym_kernel(KpmArgs k, KpmGeom kg)
{
    static_assert(NCOL >= 2, "single-colour decompositions use cheb_fast_kernel");
    using namespace ownk;
    using T = double2;
    constexpr int Q = NCOL >= 3 ? 1 : 0;
    extern __shared__ double2 lds[];
    __shared__ double red[17];
    const int N = k.N, Lt = k.Lt, Tn = blockDim.x, j = threadIdx.x;
    T *Wb[2] = {lds, lds + 2 * Tn};
    double2 *CF = lds + 4 * Tn, *CF2 = CF + k.maxorder;
    const int ncnt_ = k.sys_count > 0 ? k.sys_count : k.nsys;
    const int sys = k.sys_first + blockIdx.x % ncnt_, rank = blockIdx.x / ncnt_;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);  // heaviest orders first
    const int omc = Lt - om - 1;                                       // :523-530
    const int w = sys / k.nrhs;
    const int Lo2 = (Lt + 1) / 2;
    if (k.half && om >= Lo2) return;
    // round 1 of loads (see cheb_own_kernel)
    const bool sys_done = k.cg[sys].done != 0;  // (k.cg is never null)
    const bool act = k.active[w] != 0;
    const int n_raw = k.order[(size_t)w * k.nslot + om], n2_raw = k.order[(size_t)w * k.nslot + omc];
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    const bool on = j < kg.own_n;
    const int *own = kg.own;
    const int jq = on ? j : 0;
    const int sxq = own[jq], syq = own[Tn + jq];
    int pxq[NCOL], pyq[NCOL], cxi[NCOL], cyi[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        pxq[c] = own[(4 + 4 * c + 0) * Tn + jq];
        pyq[c] = own[(4 + 4 * c + 1) * Tn + jq];
        cxi[c] = own[(4 + 4 * c + 2) * Tn + jq];
        cyi[c] = own[(4 + 4 * c + 3) * Tn + jq];
    }
    asm volatile("" ::: "memory");  // compiler fence: keeps the loads above on this side of the early returns
    if (sys_done) return;
    const double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)om * k.nsys + sys) * N;
    double2 *prz = k.part_rz ? k.part_rz + (size_t)sys * k.rz_stride + om : nullptr;
    const int n = act ? n_raw : 1, n2 = act ? n2_raw : 1;
    const double2 *coefs = k.coefs + ((size_t)w * k.nslot + om) * k.maxorder;
    const double2 *coefs_c = k.coefs + ((size_t)w * k.nslot + omc) * k.maxorder;
    if (n <= 1) {  // single-term expansion: |c₀|² (:534)
        double f = k.scale;
        if (act) { const double2 c0 = coefs[0]; f *= c0.x * c0.x + c0.y * c0.y; }
        double acc = 0.0;
        for (int i = j; i < N; i += Tn) {
            const double2 x = v[i];
            vo[i] = make_double2(f * x.x, f * x.y);
            acc += f * (x.x * x.x + x.y * x.y);
        }
        if (prz) {
            const double t = block_sum_real(acc, red);
            if (j == 0) *prz = make_double2(t, 0.0);
        }
        return;
    }
    const double avg = 0.5 * (emax + emin), imag_ = 1.0 / (0.5 * (emax - emin));
    // round 2: gathers through the lane program
    int sx = 0, sy = 0, ox = j, oy = j;
    int px[NCOL], py[NCOL];
    double2 cx[NCOL], cy[NCOL];
    double dx = 1.0, dy = 1.0;
    const double *dbar = k.dbar + (size_t)w * N;
    const double2 *pcs = kg.pcs + (size_t)w * kg.ptotal;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) { px[c] = py[c] = j; cx[c] = cy[c] = make_double2(1.0, 0.0); }
    T ax = zero(T{}), ay = zero(T{});
    if (on) {
        sx = sxq; sy = syq;
        oy = (sy != sx) ? Tn + j : j;
        dx = dbar[sx]; dy = dbar[sy];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            px[c] = pxq[c];
            py[c] = pyq[c];
            cx[c] = pcs[cxi[c]];
            cy[c] = pcs[cyi[c]];
        }
        ax = v[sx]; ay = v[sy];
    }
    const T v0x = ax, v0y = ay;
    for (int i = j; i < n; i += Tn) CF[i] = coefs[i];      // coefficients in LDS: no global load inside the chains
    for (int i = j; i < n2; i += Tn) CF2[i] = coefs_c[i];
    __syncthreads();
    int buf = 0;
    // Σ_k cf[k] T_k(B') applied to (ax, ay), result back in (ax, ay)
    auto poly = [&](const double2 *cf, int nn) {
        if (nn <= 1) {  // not reached for mirrored frequencies of equal order; kept exact all the same
            const double2 c0 = cf[0];
            ax = cmulk(c0, ax); ay = cmulk(c0, ay);
            return;
        }
        T a1x = ax, a1y = ay, a2x = zero(T{}), a2y = zero(T{}), accx = zero(T{}), accy = zero(T{});
        for (int kk = 1; kk < nn; ++kk) {
            const double2 ck = cf[kk];
            // B̄ a = D̄ C_{L-1} … C_0 a on the lane's own two sites
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                if (c == Q) {
                    const T t_ = lin(cx[Q].x, ax, cx[Q].y, ay);
                    ay = lin(cx[Q].x, ay, cx[Q].y, ax);
                    ax = t_;
                } else {
                    T *Wc = Wb[buf];
                    buf ^= 1;
                    if (on) { Wc[ox] = ax; Wc[oy] = ay; }
                    __syncthreads();
                    const T mx = Wc[px[c]], my = Wc[py[c]];
                    ax = lin(cx[c].x, ax, cx[c].y, mx);
                    ay = lin(cy[c].x, ay, cy[c].y, my);
                }
            }
            const T xi = scl(dx, ax), xj = scl(dy, ay);
            // three-term recurrence on the lane's own sites (kpm_lmul!)
            T a3x, a3y;
            if (kk == 1) {
                a3x = scl(imag_, sub(xi, scl(avg, a1x)));
                a3y = scl(imag_, sub(xj, scl(avg, a1y)));
                const double2 c0 = cf[0];
                accx = add(cmulk(c0, a1x), cmulk(ck, a3x));
                accy = add(cmulk(c0, a1y), cmulk(ck, a3y));
            } else {
                a3x = sub(scl(imag_, scl(2.0, sub(xi, scl(avg, a2x)))), a1x);
                a3y = sub(scl(imag_, scl(2.0, sub(xj, scl(avg, a2y)))), a1y);
                accx = add(accx, cmulk(ck, a3x));
                accy = add(accy, cmulk(ck, a3y));
                a1x = a2x; a1y = a2y;
            }
            a2x = a3x; a2y = a3y;
            ax = a3x; ay = a3y;
        }
        ax = accx; ay = accy;
    };
    poly(CF2, n2);  // M̃⁻ᵀ (:526)
    poly(CF, n);    // M̃⁻¹ (:529)
    double2 acc = make_double2(0.0, 0.0);
    if (on) {
        ax = scl(k.scale, ax);
        ay = scl(k.scale, ay);
        vo[sx] = ax;
        acc.x += v0x.x * ax.x + v0x.y * ax.y;
        acc.y += v0x.x * ax.y - v0x.y * ax.x;
        if (sy != sx) {
            vo[sy] = ay;
            acc.x += v0y.x * ay.x + v0y.y * ay.y;
            acc.y += v0y.x * ay.y - v0y.y * ay.x;
        }
    }
    if (prz) {
        const double2 t = block_sum_cplx(acc, red);
        if (j == 0) *prz = t;
    }
}

// A/B switch for measurements (default on)
static int cheb_split_enabled()
{
    return tuning_env(kTuneChebSplit) == 0 ? 0 : 1;
}

static int cheb_own_enabled()
{
    return tuning_env(kTuneChebOwn) == 0 ? 0 : 1;  // A/B switch for measurements; default on
}

// ---------------------------------------------------------------------------------------------
// generic fallback (any number of colours / sites): bond tables read from memory each stage
// ---------------------------------------------------------------------------------------------
// sbari != nullptr: complex hopping means, factor [[c̄, s̄], [conj(s̄), c̄]]
__device__ __forceinline__ void bbar_colour(double2 *W, const int2 *__restrict__ bonds, const double *__restrict__ cbar, const double *__restrict__ sbar, const double *__restrict__ sbari, int cb, int ce)
{
    for (int h = cb + (int)threadIdx.x; h < ce; h += (int)blockDim.x) {
        const int2 b = bonds[h];
        const double c = cbar[h], s = sbar[h];
        const double2 a = W[b.x], d = W[b.y];
        if (sbari) {
            const double t = sbari[h];
            W[b.x] = make_double2(c * a.x + (s * d.x - t * d.y), c * a.y + (s * d.y + t * d.x));
            W[b.y] = make_double2(c * d.x + (s * a.x + t * a.y), c * d.y + (s * a.y - t * a.x));
        } else {
            W[b.x] = make_double2(c * a.x + s * d.x, c * a.y + s * d.y);
            W[b.y] = make_double2(c * d.x + s * a.x, c * d.y + s * a.y);
        }
    }
    __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void bbar_apply(double2 *W, int N, int ncol, const int2 *bonds, const int *col_off, const double *dbar, const double *cbar, const double *sbar, const double *sbari = nullptr)
{
    if (MODE == 0)
        for (int c = ncol - 1; c >= 0; --c) bbar_colour(W, bonds, cbar, sbar, sbari, col_off[c], col_off[c + 1]);
    else
        for (int c = 0; c < ncol; ++c) bbar_colour(W, bonds, cbar, sbar, sbari, col_off[c], col_off[c + 1]);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double d = (MODE == 2) ? dbar[i] * dbar[i] : dbar[i];
        W[i] = make_double2(d * W[i].x, d * W[i].y);
    }
    __syncthreads();
    if (MODE == 0)
        for (int c = 0; c < ncol; ++c) bbar_colour(W, bonds, cbar, sbar, sbari, col_off[c], col_off[c + 1]);
    else if (MODE == 2)
        for (int c = ncol - 1; c >= 0; --c) bbar_colour(W, bonds, cbar, sbar, sbari, col_off[c], col_off[c + 1]);
}

template <int MODE>
__device__ __forceinline__ void kpm_poly(double2 *W, double2 *A1, double2 *A2, double2 *ACC, const double2 *__restrict__ coefs, int n, double avg, double mag, const KpmArgs &k, const double *dbar,
                                         const double *cbar, const double *sbar, const double *sbari)
{
    const int N = k.N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) { A1[i] = ACC[i]; W[i] = ACC[i]; }
    __syncthreads();
    bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar, sbari);
    const double2 c0 = coefs[0], c1 = n > 1 ? coefs[1] : make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double2 a1 = A1[i];
        const double2 a2 = make_double2((W[i].x - avg * a1.x) / mag, (W[i].y - avg * a1.y) / mag);
        A2[i] = a2;
        W[i] = a2;
        const double2 t0 = cmulk(c0, a1), t1 = cmulk(c1, a2);
        ACC[i] = make_double2(t0.x + t1.x, t0.y + t1.y);
    }
    __syncthreads();
    for (int kk = 2; kk < n; ++kk) {
        bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar, sbari);
        const double2 ck = coefs[kk];
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double2 a1 = A1[i], a2 = A2[i];
            const double2 a3 = make_double2(2.0 * (W[i].x - avg * a2.x) / mag - a1.x, 2.0 * (W[i].y - avg * a2.y) / mag - a1.y);
            const double2 t = cmulk(ck, a3);
            ACC[i] = make_double2(ACC[i].x + t.x, ACC[i].y + t.y);
            A1[i] = a2;
            A2[i] = a3;
            W[i] = a3;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(kThreads) cheb_generic_kernel(KpmArgs k)
{
    extern __shared__ double2 lds[];
    __shared__ double red[17];
    const int N = k.N, Lt = k.Lt;
    double2 *W = k.scratch ? k.scratch + (size_t)blockIdx.x * k.scratch_stride : lds, *A1 = W + N, *A2 = A1 + N, *ACC = A2 + N;
    const int ncnt_ = k.sys_count > 0 ? k.sys_count : k.nsys;
    const int sys = k.sys_first + blockIdx.x % ncnt_, rank = blockIdx.x / ncnt_;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);
    const int w = sys / k.nrhs;
    if (k.cg[sys].done) return;  // k.cg is never null (api_handle.hip, fdm_args / kpm_args: an all-zero state outside CG loops)
    if (k.half && om >= (Lt + 1) / 2) return;
    const double *dbar = k.dbar + (size_t)w * N, *cbar = k.cbar + (size_t)w * k.Nh, *sbar = k.sbar + (size_t)w * k.Nh;
    const double *sbari = k.sbari ? k.sbari + (size_t)w * k.Nh : nullptr;
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    const double avg = 0.5 * (emax + emin), mag = 0.5 * (emax - emin);
    const double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    double2 *vo = (k.vout ? k.vout : k.v) + ((size_t)om * k.nsys + sys) * N;
    double2 *prz = k.part_rz ? k.part_rz + (size_t)sys * k.rz_stride + om : nullptr;
    const int Lo2 = (Lt + 1) / 2;
    const bool act = k.active[w] != 0;
    const int slot = k.is_sym ? (om >= Lo2 ? Lt - om - 1 : om) : om;
    const int n = act ? k.order[(size_t)w * k.nslot + slot] : 1;
    const double2 *coefs = k.coefs + ((size_t)w * k.nslot + slot) * k.maxorder;
    double2 acc = make_double2(0.0, 0.0);
    if (n > 1) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) ACC[i] = v[i];
        __syncthreads();
        if (k.is_sym) {
            kpm_poly<0>(W, A1, A2, ACC, coefs, n, avg, mag, k, dbar, cbar, sbar, sbari);
        } else {
            const int omc = Lt - om - 1;
            const double2 *coefs_c = k.coefs + ((size_t)w * k.nslot + omc) * k.maxorder;
            kpm_poly<1>(W, A1, A2, ACC, coefs_c, k.order[(size_t)w * k.nslot + omc], avg, mag, k, dbar, cbar, sbar, sbari);
            kpm_poly<1>(W, A1, A2, ACC, coefs, n, avg, mag, k, dbar, cbar, sbar, sbari);
        }
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double2 x = v[i], o = make_double2(k.scale * ACC[i].x, k.scale * ACC[i].y);
            vo[i] = o;
            acc.x += x.x * o.x + x.y * o.y;
            acc.y += x.x * o.y - x.y * o.x;
        }
    } else {
        double f = k.scale;
        if (act) f *= k.is_sym ? coefs[0].x : (coefs[0].x * coefs[0].x + coefs[0].y * coefs[0].y);
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double2 x = v[i];
            vo[i] = make_double2(f * x.x, f * x.y);
            acc.x += f * (x.x * x.x + x.y * x.y);
        }
    }
    if (prz) {
        const double2 t = block_sum_cplx(acc, red);
        if (threadIdx.x == 0) *prz = t;
    }
}

// the component-split owner-computes kernel writes TWO r·z partials per frequency (2·Lτ slots): the CG driver asks before it sizes its reductions
bool cheb_split_active(const KpmArgs &k, const KpmGeom &kg)
{
    return kg.fast && k.is_sym && k.ncol >= 2 && k.ncol <= 6 && kg.own && cheb_own_enabled() && cheb_split_enabled() && !k.half && k.sbari == nullptr;
}

// complex hoppings (round 4): Sym handles whose geometry fits the padded lists run cheb_fast_kernel<true, NCOL, CPLX>; everything else of
// a complex handle (Lanczos, Asym, the real-vector method) stays on the generic kernels (kg.fast = 0)
static bool cheb_cplx_fast(const KpmArgs &k, const KpmGeom &kg) { return !kg.fast && kg.cplx_fast && kg.pcsi && k.is_sym && k.sbari != nullptr; }

const char *cheb_kernel_name(const KpmArgs &k, const KpmGeom &kg)
{
    if (cheb_cplx_fast(k, kg)) return "cheb_fast_kernel<complex T>";
    if (!kg.fast) return "cheb_generic_kernel";
    if (cheb_wave_supported(k, kg)) return kg.wave_kind == 2 ? "cheb_wave_kernel<plaquette>" : (kg.wave_lanes == 64 ? "cheb_wave_kernel<ring, DPP>" : "cheb_wave_kernel<ring>");
    if (k.is_sym && k.ncol >= 2 && k.ncol <= 6 && kg.own && cheb_own_enabled()) return cheb_split_active(k, kg) ? (kg.wl0 && k.ncol >= 3 ? "cheb_own_kernel<split, WL0>" : "cheb_own_kernel<split>") : "cheb_own_kernel";
    if (!k.is_sym && k.ncol >= 2 && k.ncol <= 6 && kg.own && cheb_own_enabled() && k.sbari == nullptr) return "cheb_own_asym_kernel";
    return "cheb_fast_kernel";
}

void launch_cheb(hipStream_t st, const KpmArgs &k, const KpmGeom &kg)
{
    if (cheb_cplx_fast(k, kg)) {
        const size_t lds = sizeof(double2) * ((size_t)k.N + 2 * (size_t)k.maxorder);
        const int ncnt = k.sys_count > 0 ? k.sys_count : k.nsys;
        const dim3 grid((unsigned)(k.Lt * ncnt)), block((unsigned)kg.threads);
        switch (k.ncol) {
            case 1: hipLaunchKernelGGL((cheb_fast_kernel<true, 1, true>), grid, block, lds, st, k, kg); break;
            case 2: hipLaunchKernelGGL((cheb_fast_kernel<true, 2, true>), grid, block, lds, st, k, kg); break;
            case 3: hipLaunchKernelGGL((cheb_fast_kernel<true, 3, true>), grid, block, lds, st, k, kg); break;
            case 4: hipLaunchKernelGGL((cheb_fast_kernel<true, 4, true>), grid, block, lds, st, k, kg); break;
            default: hipLaunchKernelGGL((cheb_fast_kernel<true, 0, true>), grid, block, lds, st, k, kg); break;
        }
        return;
    }
    if (kg.fast) {
        const size_t lds = sizeof(double2) * ((size_t)k.N + 2 * (size_t)k.maxorder);
        const int ncnt = k.sys_count > 0 ? k.sys_count : k.nsys;
        const dim3 grid((unsigned)(k.Lt * ncnt)), block((unsigned)kg.threads);
#define CHEB_LAUNCH(S_, C_) hipLaunchKernelGGL((cheb_fast_kernel<S_, C_>), grid, block, lds, st, k, kg)
        if (cheb_wave_supported(k, kg)) {
            launch_cheb_wave(st, k, kg);  // ring / plaquette lattices: one wavefront per chain, no LDS exchange, no barrier (kernels_kpm_wave.hip)
        } else if (k.is_sym && k.ncol >= 2 && k.ncol <= 6 && kg.own && cheb_own_enabled()) {
            const bool split = cheb_split_active(k, kg);
            // heavy / light workgroups (see cheb_own_kernel): k.group > 0 says that k.heavy is the host's count of leading frequencies with
            // more than one term (possibly 0); otherwise every frequency gets its own workgroup(s).  SMOQY_CHEB_GROUP=1 for A/B runs.
            static const int env_group = tuning_env(kTuneChebGroup) > 0 ? tuning_env(kTuneChebGroup) : 0;
            KpmArgs kk = k;
            kk.group = std::min(8, env_group > 0 ? env_group : k.group);  // light workgroups hold at most 8 frequencies in registers
            if (kk.group <= 1) { kk.group = 1; kk.heavy = k.Lt; }
            kk.heavy = std::max(0, std::min(k.Lt, kk.heavy));
            const int nlight = (k.Lt - kk.heavy + kk.group - 1) / kk.group;
            const size_t olds = (split ? sizeof(double) : sizeof(double2)) * 4 * (size_t)kg.threads + sizeof(double2) * (size_t)k.maxorder;
            const dim3 ogrid((unsigned)(((split ? 2 : 1) * kk.heavy + nlight) * ncnt));
            static const int env_wl = tuning_env(kTuneChebWl0) == 0 ? 0 : 1;  // A/B switch, default on
            // wave-local colour-0 exchange: 0 off, 1 ds_bpermute, 2 / 3 DPP row rotations (KpmGeom::wl0; SMOQY_CHEB_WL0=0 / 1 caps it)
            static const int env_wl_cap = tuning_env(kTuneChebWl0) >= 0 ? tuning_env(kTuneChebWl0) : 3;
            int wl0 = (k.ncol >= 3 && env_wl) ? kg.wl0 : 0;
            if (wl0 > 1 && (env_wl_cap == 1 || !split)) wl0 = 1;  // the DPP forms are instantiated for the component-split kernel only
#define OWN_LAUNCH(C_)                                                                                          \
    {                                                                                                           \
        if (split && wl0 == 2) hipLaunchKernelGGL((cheb_own_kernel<C_, true, 2>), ogrid, block, olds, st, kk, kg); \
        else if (split && wl0 == 3) hipLaunchKernelGGL((cheb_own_kernel<C_, true, 3>), ogrid, block, olds, st, kk, kg); \
        else if (split && wl0) hipLaunchKernelGGL((cheb_own_kernel<C_, true, 1>), ogrid, block, olds, st, kk, kg); \
        else if (split) hipLaunchKernelGGL((cheb_own_kernel<C_, true>), ogrid, block, olds, st, kk, kg);         \
        else if (wl0) hipLaunchKernelGGL((cheb_own_kernel<C_, false, 1>), ogrid, block, olds, st, kk, kg);       \
        else hipLaunchKernelGGL((cheb_own_kernel<C_, false>), ogrid, block, olds, st, kk, kg);                   \
    }
            switch (k.ncol) {
                case 2: OWN_LAUNCH(2); break;
                case 3: OWN_LAUNCH(3); break;
                case 4: OWN_LAUNCH(4); break;
                case 5: OWN_LAUNCH(5); break;
                default: OWN_LAUNCH(6); break;
            }
#undef OWN_LAUNCH
        } else if (k.is_sym) {
            switch (k.ncol) {
                case 1: CHEB_LAUNCH(true, 1); break;
                case 2: CHEB_LAUNCH(true, 2); break;
                case 3: CHEB_LAUNCH(true, 3); break;
                case 4: CHEB_LAUNCH(true, 4); break;
                default: CHEB_LAUNCH(true, 0); break;
            }
        } else if (k.ncol >= 2 && k.ncol <= 6 && kg.own && cheb_own_enabled() && k.sbari == nullptr) {
            const size_t olds = sizeof(double2) * (4 * (size_t)kg.threads + 2 * (size_t)k.maxorder);
            switch (k.ncol) {
                case 2: hipLaunchKernelGGL((cheb_own_asym_kernel<2>), grid, block, olds, st, k, kg); break;
                case 3: hipLaunchKernelGGL((cheb_own_asym_kernel<3>), grid, block, olds, st, k, kg); break;
                case 4: hipLaunchKernelGGL((cheb_own_asym_kernel<4>), grid, block, olds, st, k, kg); break;
                case 5: hipLaunchKernelGGL((cheb_own_asym_kernel<5>), grid, block, olds, st, k, kg); break;
                default: hipLaunchKernelGGL((cheb_own_asym_kernel<6>), grid, block, olds, st, k, kg); break;
            }
        } else {
            switch (k.ncol) {
                case 1: CHEB_LAUNCH(false, 1); break;
                case 2: CHEB_LAUNCH(false, 2); break;
                case 3: CHEB_LAUNCH(false, 3); break;
                case 4: CHEB_LAUNCH(false, 4); break;
                default: CHEB_LAUNCH(false, 0); break;
            }
        }
#undef CHEB_LAUNCH
    } else {
        const size_t lds = k.scratch ? 0 : sizeof(double2) * 4 * (size_t)k.N;
        hipLaunchKernelGGL(cheb_generic_kernel, dim3((unsigned)(k.Lt * (k.sys_count > 0 ? k.sys_count : k.nsys))), dim3(kThreads), lds, st, k);
    }
}

__global__ void conj_mirror_kernel(double2 *v, int Lt, int N, int nsys)
{
    const size_t per = (size_t)nsys * N;
    const int Lo2 = (Lt + 1) / 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per * Lo2; idx += (size_t)gridDim.x * blockDim.x) {
        const int om = (int)(idx / per);
        const size_t r = idx - (size_t)om * per;
        const double2 x = v[(size_t)om * per + r];
        v[(size_t)(Lt - 1 - om) * per + r] = make_double2(x.x, -x.y);
    }
}

void launch_conj_mirror(hipStream_t st, double2 *v, int Lt, int N, int nsys)
{
    const size_t tot = (size_t)nsys * N * ((Lt + 1) / 2);
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conj_mirror_kernel, dim3(blocks), dim3(256), 0, st, v, Lt, N, nsys);
}

// ---------------------------------------------------------------------------------------------
// device-side bookkeeping of update_preconditioner! (KPMPreconditioner.jl:565-597, 696-731): runs at the end of the Lanczos kernel,
// in the same workgroup (one per walker)
// ---------------------------------------------------------------------------------------------
// number of eigenvalues of the symmetric tridiagonal (a, b) below x: sign changes of the leading principal minors p_i(x) of T − x
// (p_0 = 1, p_1 = a_0 − x, p_{i+1} = (a_i − x) p_i − b_{i-1}² p_{i-1}).  Round 4: the quotient form q_i = p_i / p_{i-1} this replaces put one fp64
// DIVISION (77 cycles, tools/phase_probe.hip) on the dependent chain of every one of the n steps of every one of the 11 rounds of the
// 64-section below — ~17 of the 44 us of a 20-step Lanczos launch; here the chain is one multiply-add per step.  A zero minor takes the sign
// of its predecessor (the quotient form's "q = 0 is not negative"); the pair is rescaled by an exact power of two before it can leave the
// double range (n_lanczos may be as large as 1024).
__device__ __forceinline__ int sturm_count_dev(const double *a, const double *b, int n, double x)
{
    double p0 = 1.0, p1 = a[0] - x;
    bool s1 = p1 < 0.0;  // sign of p_1 against p_0 = 1
    int cnt = s1 ? 1 : 0;
    int i = 1;
    auto step = [&](double ai, double bi) {
        const double p2 = (ai - x) * p1 - (bi * bi) * p0;
        const bool s2 = p2 != 0.0 ? (p2 < 0.0) : s1;
        cnt += (s2 != s1) ? 1 : 0;
        p0 = p1; p1 = p2; s1 = s2;
    };
    for (; i + 3 < n; i += 4) {
        const double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];   // the loads of a block issue together
        const double b0 = b[i - 1], b1 = b[i], b2 = b[i + 1], b3 = b[i + 2];
        step(a0, b0); step(a1, b1); step(a2, b2); step(a3, b3);
        if (fabs(p1) > 0x1p+400 || fabs(p0) > 0x1p+400) { p0 *= 0x1p-400; p1 *= 0x1p-400; }
        else if (fabs(p1) < 0x1p-400 && fabs(p0) < 0x1p-400) { p0 *= 0x1p+400; p1 *= 0x1p+400; }
    }
    for (; i < n; ++i) step(a[i], b[i - 1]);
    return cnt;
}

// smallest (target = 1) or largest (target = n) eigenvalue by 64-section inside one wavefront: every round the 64 lanes evaluate the
// Sturm count at 64 interior points of [l, h] and a ballot picks the sub-interval where the count crosses `target`; eleven rounds
// shrink the Gershgorin interval by 65^11 > 2^64 (the host form did 64 halvings).  All lanes return the same value.
__device__ __forceinline__ double tridiag_extreme_wave(const double *a, const double *b, int n, double lo, double hi, int target)
{
    const int lane = threadIdx.x & 63;
    double l = lo, h = hi;
    for (int round = 0; round < 11; ++round) {
        const double step = (h - l) / 65.0;
        const double x = l + step * (double)(lane + 1);
        const unsigned long long m = __ballot(sturm_count_dev(a, b, n, x) >= target);
        if (m == 0ull) {
            l = l + step * 64.0;
        } else {
            const int f = __ffsll((long long)m) - 1;  // first point at which the count has reached the target
            h = l + step * (double)(f + 1);
            l = l + step * (double)f;
        }
    }
    return 0.5 * (l + h);
}

// a, b: the Lanczos coefficients of this workgroup's walker (n and n - 1 of them; LDS or global memory, written before the last barrier).
// sh: >= 4 doubles + 4 ints of LDS scratch.  Every lane of the workgroup must call it.
__device__ __forceinline__ void precond_bookkeeping(const double *a, const double *b, int n, int w, const PreUpd &u, double *sh)
{
    int *shi_ = reinterpret_cast<int *>(sh + 4);
    const int wave = threadIdx.x >> 6, nwave = (blockDim.x + 63) >> 6;
    if (threadIdx.x == 0) { shi_[0] = 0; shi_[1] = 0; }
    if (wave < 2) {
        // Gershgorin interval (every lane the same arithmetic), then wave 0 takes the smallest and wave 1 (or wave 0 again) the largest eigenvalue
        double lo = a[0], hi = a[0];
        for (int i = 0; i < n; ++i) {
            const double r = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i < n - 1 ? fabs(b[i]) : 0.0);
            lo = fmin(lo, a[i] - r);
            hi = fmax(hi, a[i] + r);
        }
        if (wave == 0) {
            const double e = tridiag_extreme_wave(a, b, n, lo, hi, 1);
            if (threadIdx.x == 0) sh[0] = e;
        }
        if (wave == (nwave > 1 ? 1 : 0)) {
            const double e = tridiag_extreme_wave(a, b, n, lo, hi, n);
            if ((threadIdx.x & 63) == 0) sh[1] = e;
        }
    }
    __syncthreads();
    double emin = sh[0], emax = sh[1];
    if (!u.is_sym) { emin = sqrt(emin); emax = sqrt(emax); }  // Lanczos ran on B̄ᵀB̄ (:655)
    emin *= (1.0 - u.rbuf);                                   // :569-570
    emax *= (1.0 + u.rbuf);
    const double oe = u.bounds[2 * w], oE = u.bounds[2 * w + 1];
    const bool ok = 0.0 < emin && emin < 1.0 && 1.0 < emax && emax < 2.0;                                       // :573
    const bool moved = ok && (fabs((emin - oe) / oe) > u.rbuf / 2 || fabs((emax - oE) / oE) > u.rbuf / 2);    // :582 (first update: oe = 0, the quotient is +inf)
    int nrebuild = u.status[4 * w];
    __syncthreads();  // every lane has read the old bounds before lane 0 replaces them
    if (moved) {
        // update_kpm_expansion_order! (:696-731)
        int mx = 1, last = -1;
        for (int l = threadIdx.x; l < u.nslot; l += blockDim.x) {
            double phi = 2.0 * M_PI / u.Lt * (l + 0.5);  // :220
            if (phi > M_PI) phi = 2.0 * M_PI - phi;       // :710
            int o = (int)floor((emax - emin) * (u.a1 / phi + u.a2));  // :711
            o = max(o, 1);
            o = min(o, u.maxorder);  // cannot bind: the table is sized for emax - emin < 2, which `ok` guarantees
            u.order[(size_t)w * u.nslot + l] = o;
            mx = max(mx, o);
            if (o > 1) last = max(last, u.is_sym ? l : min(l, u.Lt - 1 - l));
        }
        atomicMax(&shi_[0], mx);
        atomicMax(&shi_[1], last + 1);
        __syncthreads();
        if (threadIdx.x == 0) {
            u.bounds[2 * w] = emin;
            u.bounds[2 * w + 1] = emax;
            u.status[4 * w] = nrebuild + 1;
            u.status[4 * w + 1] = min(u.Lt, 2 * shi_[1]);  // ranks 2s and 2s+1 share slot s (:387): the heavy frequencies of cheb_own_kernel
            u.status[4 * w + 2] = shi_[0];
        }
    }
    if (threadIdx.x == 0) {
        u.active[w] = ok ? 1 : 0;       // :575 / :593
        u.rebuild[w] = moved ? 1 : 0;
        u.status[4 * w + 3] = ok ? 1 : 0;
    }
}

// update_kpm_expansion_coefs! (:734-795) with kpm_coefs! restated (SmoQyKPMCore: Chebyshev-Gauss quadrature with 2n nodes, no damping
// kernel): workgroup (slot l < cld(Lτ, 2), walker), one wavefront.  coefs[k] = (2 - δ_k0)/M Σ_j f(x_j) cos(π k (j + ½)/M), M = 2n,
// x_j = avg + mag cos(π (j + ½)/M), summed over j in the order of the host restatement it replaces.
__global__ void __launch_bounds__(64) kpm_expansions_kernel(PreUpd u, int w0)
{
    extern __shared__ double gl[];  // [2][M]
    const int w = w0 + blockIdx.y, l = blockIdx.x;
    if (!u.rebuild[w]) return;
    const int n = u.order[(size_t)w * u.nslot + l], M = 2 * n;
    const double emin = u.bounds[2 * w], emax = u.bounds[2 * w + 1];
    const double avg = 0.5 * (emax + emin), mag = 0.5 * (emax - emin);
    const double phi = 2.0 * M_PI / u.Lt * (l + 0.5);
    const double cp = cos(phi), sp = sin(phi);
    double *gre = gl, *gim = gl + M;
    const double inv2M = 1.0 / (2.0 * M);
    for (int j = threadIdx.x; j < M; j += 64) {
        const double bx = avg + mag * cospi((double)(2 * j + 1) * inv2M);
        if (u.is_sym) {
            gre[j] = 1.0 / (bx * bx - 2.0 * bx * cp + 1.0);  // f_B̄_sym :800
        } else {
            const double x = 1.0 - bx * cp, y = bx * sp;      // f_B̄_asym = 1/(1 - e^{-iφ} b) :804
            gre[j] = x / (x * x + y * y);
            gim[j] = -y / (x * x + y * y);
        }
    }
    __syncthreads();
    double2 *out = u.coefs + ((size_t)w * u.nslot + l) * u.maxorder;
    double2 *outc = u.is_sym ? nullptr : u.coefs + ((size_t)w * u.nslot + (u.Lt - l - 1)) * u.maxorder;
    for (int k = threadIdx.x; k < n; k += 64) {
        double are = 0.0, aim = 0.0;
        for (int j = 0; j < M; ++j) {
            const int m = (int)(((long long)k * (2 * j + 1)) % (4 * M));
            const double cc = cospi((double)m * inv2M);
            are += gre[j] * cc;
            if (!u.is_sym) aim += gim[j] * cc;
        }
        const double f = (k == 0 ? 1.0 : 2.0) / M;
        out[k] = make_double2(f * are, f * aim);
        if (outc) outc[k] = make_double2(f * are, -f * aim);  // :791 (the middle frequency of an odd Lτ mirrors onto itself and ends up conjugated, as in the reference)
    }
}

void launch_kpm_expansions(hipStream_t st, const PreUpd &u, int w0, int nw)
{
    const int Lo2 = (u.Lt + 1) / 2;
    const size_t lds = sizeof(double) * 4 * (size_t)u.maxorder;
    hipLaunchKernelGGL(kpm_expansions_kernel, dim3((unsigned)Lo2, (unsigned)nw), dim3(64), lds, st, u, w0);
}

// ---------------------------------------------------------------------------------------------
// lanczos! (SmoQyKPMCore, restated): n-step Lanczos on B̄ (Sym) or B̄ᵀB̄ (Asym) from the host
// supplied start vectors; one workgroup per walker, everything in LDS / registers.
// ---------------------------------------------------------------------------------------------
template <int MODE, bool FAST>
__global__ void __launch_bounds__(1024) lanczos_kernel(KpmArgs k, KpmGeom kg, int w0, const double *__restrict__ randvec, int nsteps, double *alpha, double *beta, PreUpd u)
{
    extern __shared__ double2 lds[];
    __shared__ double red[17];
    const int N = k.N, w = w0 + blockIdx.x;
    double2 *W = (!FAST && k.scratch) ? k.scratch + (size_t)blockIdx.x * k.scratch_stride : lds, *VK = W + N, *VKM = VK + N;
    const double *dbar = k.dbar + (size_t)w * N, *cbar = k.cbar + (size_t)w * k.Nh, *sbar = k.sbar + (size_t)w * k.Nh;
    // T = ComplexF64 (generic path only): the start vector is N complex deviates (randn! on a Vector{ComplexF64},
    // KPMPreconditioner.jl:634) and B̄ is complex Hermitian; the Lanczos vectors are then genuinely complex.  For real hoppings every
    // imaginary part below is an exact zero and the arithmetic reduces bit for bit to the real recurrence.
    const double *sbari = (!FAST && k.sbari) ? k.sbari + (size_t)w * k.Nh : nullptr;
    const bool cplx = sbari != nullptr;
    randvec += (size_t)blockIdx.x * N * (cplx ? 2 : 1);
    alpha += (size_t)blockIdx.x * 1024;
    beta += (size_t)blockIdx.x * 1024;
    LaneBonds lb;
    if (FAST) load_lane_bonds(lb, kg, w, k.ncol);
    double acc = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc += cplx ? randvec[2 * i] * randvec[2 * i] + randvec[2 * i + 1] * randvec[2 * i + 1] : randvec[i] * randvec[i];
    const double nrm = sqrt(block_sum_real(acc, red));
    // the fast path works in LDS-position order (a permutation: dot products are unaffected)
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        VK[FAST ? kg.pos[i] : i] = cplx ? make_double2(randvec[2 * i] / nrm, randvec[2 * i + 1] / nrm) : make_double2(randvec[i] / nrm, 0.0);
        VKM[i] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    double bprev = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) W[i] = VK[i];
        __syncthreads();
        if (FAST) bbar_apply_regs<MODE>(W, lb, k.ncol, N, dbar, kg.pos);
        else bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar, sbari);
        acc = 0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) acc += FAST ? VK[i].x * W[i].x : VK[i].x * W[i].x + VK[i].y * W[i].y;
        const double al = block_sum_real(acc, red);
        acc = 0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double wv = W[i].x - al * VK[i].x - bprev * VKM[i].x;
            const double wi = FAST ? 0.0 : W[i].y - al * VK[i].y - bprev * VKM[i].y;
            W[i] = make_double2(wv, wi);
            acc += FAST ? wv * wv : wv * wv + wi * wi;
        }
        const double nb = sqrt(block_sum_real(acc, red));
        if (threadIdx.x == 0) {
            alpha[s] = al;
            if (s < nsteps - 1) beta[s] = nb;
        }
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            VKM[i] = VK[i];
            VK[i] = make_double2(W[i].x / nb, FAST ? 0.0 : W[i].y / nb);
        }
        bprev = nb;
        __syncthreads();
    }
    // lane 0's stores of alpha / beta are visible to the workgroup behind the loop's last barrier
    precond_bookkeeping(alpha, beta, nsteps, w, u, red);
}

// Owner-computes Lanczos (Sym, real hoppings, 2..kMaxColours colours): the lane keeps its two sites of v_k, v_{k-1} and w in registers,
// B̄ is applied through the lane program of cheb_own_kernel (three exchanges per apply on the honeycomb lattice instead of five LDS
// read-modify-write stages) and each of the two inner products of a step costs one barrier (wave sums ping-pong between two LDS rows).
// Arithmetic per site is that of lanczos_kernel; the inner products add the sites in lane order instead of LDS-position order.
template <int NCOL, int WL0>
__global__ void __launch_bounds__(1024) lanczos_own_kernel(KpmArgs k, KpmGeom kg, int w0, const double *__restrict__ randvec, int nsteps, double *alpha, double *beta, PreUpd u)
{
    using namespace ownk;
    using T = double;
    extern __shared__ double2 lds[];
    __shared__ double red2[2][16];
    __shared__ double sh[8];
    const int N = k.N, w = w0 + blockIdx.x, Tn = blockDim.x, j = threadIdx.x;
    T *Wb0 = reinterpret_cast<T *>(lds), *Wb1 = Wb0 + 2 * Tn;
    double *sa = Wb1 + 2 * Tn, *sb = sa + nsteps;  // the Lanczos coefficients stay in LDS for the bookkeeping at the end
    randvec += (size_t)blockIdx.x * N;
    alpha += (size_t)blockIdx.x * 1024;
    beta += (size_t)blockIdx.x * 1024;
    OwnIdx<NCOL> I;
    own_load_idx<NCOL>(I, kg, Tn, j);
    OwnProg<NCOL> P;
    own_load_prog<NCOL>(P, I, k.dbar + (size_t)w * N, kg.pcs + (size_t)w * kg.ptotal, Tn, j);
    const bool two = P.on && P.sy != P.sx;
    const int wave = j >> 6, lane = j & 63, nwave = (Tn + 63) >> 6;
    int rb = 0;
    if (j < 32) red2[j >> 4][j & 15] = 0.0;  // rows past the last wavefront stay zero: the sum below reads a fixed number of slots
    __syncthreads();
    auto bsum = [&](double v) {  // workgroup sum, one barrier: safe because consecutive calls alternate rows and a barrier lies between two uses of a row
        v = wsum_k(v);
        if (lane == 0) red2[rb][wave] = v;
        __syncthreads();
        // the wave sums are added in wave order, as before; reading a FIXED count (4 or 16 slots, zeros behind the last wavefront: x + 0 = x)
        // lets the reads issue together — the loop over the run-time wave count compiled to one LDS round trip per wave sum, ~480 cycles of
        // the ~2600 of a Lanczos step, twice per step (round 4, tools/phase_probe.hip for the primitives)
        const double *r = red2[rb];
        double t = 0.0;
        if (nwave <= 4) {
            const double r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            t = (((t + r0) + r1) + r2) + r3;
        } else {
            double rr[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) rr[q] = r[q];
#pragma unroll
            for (int q = 0; q < 16; ++q) t += rr[q];
        }
        rb ^= 1;
        return t;
    };
    double vkx = P.on ? randvec[P.sx] : 0.0, vky = two ? randvec[P.sy] : 0.0;
    const double nrm = sqrt(bsum(vkx * vkx + vky * vky));
    vkx /= nrm; vky /= nrm;
    double vmx = 0.0, vmy = 0.0, bprev = 0.0;
    int buf = 0;
    for (int s = 0; s < nsteps; ++s) {
        T ax = vkx, ay = two ? vky : vkx;  // a self bond's lane carries its one site in both slots
        own_bbar_apply<NCOL, T, WL0>(P, ax, ay, Wb0, Wb1, buf);
        const double al = bsum(P.on ? vkx * ax + (two ? vky * ay : 0.0) : 0.0);
        const double wx = ax - al * vkx - bprev * vmx;
        const double wy = two ? ay - al * vky - bprev * vmy : 0.0;
        const double nb = sqrt(bsum(P.on ? wx * wx + wy * wy : 0.0));
        if (j == 0) {  // LDS only inside the loop: a global store here is waited for (vmcnt) at the next barrier of every step
            sa[s] = al;
            if (s < nsteps - 1) sb[s] = nb;
        }
        vmx = vkx; vmy = vky;
        vkx = wx / nb; vky = wy / nb;
        bprev = nb;
    }
    __syncthreads();
    for (int q = j; q < nsteps; q += Tn) {   // smoqy_precond_get reads them back
        alpha[q] = sa[q];
        if (q < nsteps - 1) beta[q] = sb[q];
    }
    precond_bookkeeping(sa, sb, nsteps, w, u, sh);
}

// raise the dynamic-LDS limit of the generic kernels once, outside any stream capture
hipError_t configure_kpm_kernels(const char **what)
{
    hipError_t first = hipSuccess;
    SMOQY_SET_LDS(cheb_generic_kernel, 160 * 1024 - 256);
    SMOQY_SET_LDS((lanczos_kernel<0, false>), 160 * 1024 - 256);
    SMOQY_SET_LDS((lanczos_kernel<2, false>), 160 * 1024 - 256);
    SMOQY_SET_LDS(kpm_expansions_kernel, 160 * 1024 - 256);
#define SMOQY_LANCZOS_OWN_LDS(C_)                                         \
    SMOQY_SET_LDS((lanczos_own_kernel<C_, 0>), 160 * 1024 - 1024); /* 320 bytes of static LDS besides */ \
    SMOQY_SET_LDS((lanczos_own_kernel<C_, 1>), 160 * 1024 - 1024);         \
    SMOQY_SET_LDS((lanczos_own_kernel<C_, 2>), 160 * 1024 - 1024);         \
    SMOQY_SET_LDS((lanczos_own_kernel<C_, 3>), 160 * 1024 - 1024);
    SMOQY_LANCZOS_OWN_LDS(2)
    SMOQY_LANCZOS_OWN_LDS(3)
    SMOQY_LANCZOS_OWN_LDS(4)
    SMOQY_LANCZOS_OWN_LDS(5)
    SMOQY_LANCZOS_OWN_LDS(6)
#undef SMOQY_LANCZOS_OWN_LDS
    // owner-computes Chebyshev kernels: four images of `threads` values + the coefficient tables pass 64 KB at 1024 threads (complex values)
#define SMOQY_OWN_LDS(C_)                                                  \
    SMOQY_SET_LDS((cheb_own_kernel<C_, false>), 160 * 1024 - 256);         \
    SMOQY_SET_LDS((cheb_own_kernel<C_, true>), 160 * 1024 - 256);          \
    SMOQY_SET_LDS((cheb_own_kernel<C_, false, 1>), 160 * 1024 - 256);      \
    SMOQY_SET_LDS((cheb_own_kernel<C_, true, 1>), 160 * 1024 - 256);       \
    SMOQY_SET_LDS((cheb_own_kernel<C_, true, 2>), 160 * 1024 - 256);       \
    SMOQY_SET_LDS((cheb_own_kernel<C_, true, 3>), 160 * 1024 - 256);       \
    SMOQY_SET_LDS((cheb_own_asym_kernel<C_>), 160 * 1024 - 256);
    SMOQY_OWN_LDS(2) SMOQY_OWN_LDS(3) SMOQY_OWN_LDS(4) SMOQY_OWN_LDS(5) SMOQY_OWN_LDS(6)
#undef SMOQY_OWN_LDS
    return first;
}

void launch_lanczos(hipStream_t st, const KpmArgs &k, const KpmGeom &kg, int w0, int nw, const double *randvec, int nsteps, double *alpha, double *beta, bool use_BtB, const PreUpd &u)
{
    const size_t lds = (!kg.fast && k.scratch) ? 0 : sizeof(double2) * 3 * (size_t)k.N;
    const int threads = kg.fast ? kg.threads : kThreads;
    if (kg.fast && !use_BtB && k.ncol >= 2 && k.ncol <= kMaxColours && kg.own && cheb_own_enabled() && k.sbari == nullptr) {
        const size_t olds = sizeof(double) * (4 * (size_t)kg.threads + 2 * (size_t)nsteps);
        const int wl0 = k.ncol >= 3 ? kg.wl0 : 0;
#define LANCZOS_OWN(C_)                                                                                                                         \
    {                                                                                                                                           \
        if (wl0 == 2) hipLaunchKernelGGL((lanczos_own_kernel<C_, 2>), dim3(nw), dim3(threads), olds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);  \
        else if (wl0 == 3) hipLaunchKernelGGL((lanczos_own_kernel<C_, 3>), dim3(nw), dim3(threads), olds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);  \
        else if (wl0) hipLaunchKernelGGL((lanczos_own_kernel<C_, 1>), dim3(nw), dim3(threads), olds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);  \
        else hipLaunchKernelGGL((lanczos_own_kernel<C_, 0>), dim3(nw), dim3(threads), olds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);     \
    }
        switch (k.ncol) {
            case 2: LANCZOS_OWN(2); break;
            case 3: LANCZOS_OWN(3); break;
            case 4: LANCZOS_OWN(4); break;
            case 5: LANCZOS_OWN(5); break;
            default: LANCZOS_OWN(6); break;
        }
#undef LANCZOS_OWN
        return;
    }
    if (kg.fast) {
        if (use_BtB) hipLaunchKernelGGL((lanczos_kernel<2, true>), dim3(nw), dim3(threads), lds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);
        else hipLaunchKernelGGL((lanczos_kernel<0, true>), dim3(nw), dim3(threads), lds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);
    } else {
        if (use_BtB) hipLaunchKernelGGL((lanczos_kernel<2, false>), dim3(nw), dim3(threads), lds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);
        else hipLaunchKernelGGL((lanczos_kernel<0, false>), dim3(nw), dim3(threads), lds, st, k, kg, w0, randvec, nsteps, alpha, beta, u);
    }
}

}  // namespace smoqy
