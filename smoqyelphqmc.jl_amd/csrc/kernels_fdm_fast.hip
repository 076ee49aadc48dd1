// Register-resident FermionDetMatrix kernels for gfx950 (Sym, <= kFdmColours colours, tau-chunk
// <= 2): M, Mᵀ, MᵀM, MMᵀ as ONE launch each.  Same semantics as kernels_fdm.hip (which stays as the
// generic path for Asym / many colours / large chunks); reference: src/FermionDetMatrix.jl:385-427,
// 484-525, 329-340, 357-368 and src/checkerboard_matrix_multiply.jl:50-69.
//
// What changed relative to the generic kernel and why:
//   * every colour's bond list is padded with identity self bonds so that it covers all N sites
//     and a lane owns exactly one (padded) bond per colour; the cosh/sinh pairs are stored
//     interleaved per padded bond (csf[w][l][idx] = (c, s)), so a lane fetches ALL the hopping
//     data it will ever need with one coalesced 16-byte load per (colour, slice) at kernel entry;
//     after that the stage chain touches LDS only — no global-load latency between barriers;
//   * exp(-ΔτV) is folded into the first colour's stage (C₁ D C₁ in registers): 2L-1 stages;
//   * the last colour's stage leaves B v in the registers of the lane owning that bond; the
//     "v ∓ B v" combines, the hand-over from M to Mᵀ inside the fused MᵀM, the output store and
//     the dot(in, out) partial all happen on those registers — the intermediate M v never exists
//     in memory and only one LDS array (the propagating slices) is needed.
#include "smoqy_internal.h"

#include <cstdlib>

namespace smoqy {

namespace {

constexpr int KMAX = 3;  // slices a workgroup propagates at once (tau-chunk + 1 halo)

__device__ __forceinline__ int wrapl(int l, int Lt) { return l >= Lt ? l - Lt : (l < 0 ? l + Lt : l); }
__device__ __forceinline__ double2 lin(double a, double2 x, double b, double2 y) { return make_double2(a * x.x + b * y.x, a * x.y + b * y.y); }
__device__ __forceinline__ double2 scl(double a, double2 x) { return make_double2(a * x.x, a * x.y); }
// v - ph·u with the hop phase ph = hop (M rows) or conj(hop) (Mᵀ rows), negated on the wrap-around row
// when the time direction is antiperiodic.  hop = 1, antiperiodic = 1 is the reference operator
// (v - u, v + u on the wrap row, bit for bit); the CG runs with hop = exp(-iπ/Lτ), periodic.
__device__ __forceinline__ double2 hopcomb(double2 v, double2 u, bool wrap, bool dagger, const FdmArgs &a)
{
    double pr = a.hop_re, pi = dagger ? -a.hop_im : a.hop_im;
    if (wrap && a.antiperiodic) { pr = -pr; pi = -pi; }
    return make_double2(v.x - (pr * u.x - pi * u.y), v.y - (pr * u.y + pi * u.x));
}

// CSV = false: the host has shown the hoppings to be τ-independent (FdmFast::cs_const), one (cosh, sinh) pair per colour is kept
// instead of one per slice — 12 fewer registers per colour and no copies
template <bool CSV = true, bool CPLX = false>
struct LaneT {
    static constexpr bool kCplx = CPLX;
    int2 b[kFdmColours];
    bool on[kFdmColours];
    double2 cs[kFdmColours][CSV ? KMAX : 1];  // (cosh, Re sinh) of the lane's bond in colour c on slice k
    double si[CPLX ? kFdmColours : 1][CPLX ? (CSV ? KMAX : 1) : 1];  // T = ComplexF64: Im sinh (round 4)
    double di[KMAX], dj[KMAX];                // exp(-ΔτV) at the two sites of the lane's first-colour bond
    __device__ __forceinline__ double2 csk(int c, int k) const { return cs[c][CSV ? k : 0]; }
    __device__ __forceinline__ double sik(int c, int k) const { return CPLX ? si[CPLX ? c : 0][CPLX && CSV ? k : 0] : 0.0; }
};
// the bond factor [[c, s], [conj(s), c]] on the pair (a, d) = (u_i, u_j) of a bond (i, j) in the order of the neighbour table
// (src/checkerboard_matrix_multiply.jl:60-68): a' = c a + s d, d' = c d + conj(s) a.  Real hoppings: si = 0 and the two extra terms fold away.
template <bool CPLX>
__device__ __forceinline__ void bond_apply(double c, double sr, double si, double2 a, double2 d, double2 &oa, double2 &od)
{
    oa = lin(c, a, sr, d);
    od = lin(c, d, sr, a);
    if (CPLX) {
        oa.x -= si * d.y; oa.y += si * d.x;
        od.x += si * a.y; od.y -= si * a.x;
    }
}
using Lane = LaneT<true>;

// one plain colour stage on nk LDS-resident slices; SH selects the slice->register offset
template <int C, int SH, class LN>
__device__ __forceinline__ void stage(double2 *U, int N, int nk, const LN &ln)
{
    if (ln.on[C]) {
#pragma unroll
        for (int k = 0; k < KMAX - SH; ++k) {
            if (k < nk) {
                double2 *row = U + (size_t)k * N;
                const double2 a = row[ln.b[C].x], d = row[ln.b[C].y];
                const double2 cs_ = ln.csk(C, k + SH);
                double2 oa, od;
                bond_apply<LN::kCplx>(cs_.x, cs_.y, ln.sik(C, k + SH), a, d, oa, od);
                row[ln.b[C].x] = oa;
                row[ln.b[C].y] = od;
            }
        }
    }
    __syncthreads();
}

// first colour with the diagonal folded in: C₁ D C₁.  LAST: keep the result in registers.
template <int SH, bool LAST, class LN>
__device__ __forceinline__ void middle(double2 *U, int N, int nk, const LN &ln, double2 (&ri)[KMAX], double2 (&rj)[KMAX])
{
    if (ln.on[0]) {
#pragma unroll
        for (int k = 0; k < KMAX - SH; ++k) {
            if (k < nk) {
                double2 *row = U + (size_t)k * N;
                const double2 a = row[ln.b[0].x], d = row[ln.b[0].y];
                const double2 cs_ = ln.csk(0, k + SH);
                const double c = cs_.x, s = cs_.y, si = ln.sik(0, k + SH);
                double2 x, y, ox, oy;
                bond_apply<LN::kCplx>(c, s, si, a, d, x, y);
                x = scl(ln.di[k + SH], x);
                y = scl(ln.dj[k + SH], y);
                bond_apply<LN::kCplx>(c, s, si, x, y, ox, oy);
                if (LAST) {
                    ri[k] = ox;
                    rj[k] = oy;
                } else {
                    row[ln.b[0].x] = ox;
                    row[ln.b[0].y] = oy;
                }
            }
        }
    }
    if (!LAST) __syncthreads();
}

// last colour: results stay in registers (no LDS write, no barrier)
template <int C, int SH, class LN>
__device__ __forceinline__ void last_stage(const double2 *U, int N, int nk, const LN &ln, double2 (&ri)[KMAX], double2 (&rj)[KMAX])
{
    if (ln.on[C]) {
#pragma unroll
        for (int k = 0; k < KMAX - SH; ++k) {
            if (k < nk) {
                const double2 *row = U + (size_t)k * N;
                const double2 a = row[ln.b[C].x], d = row[ln.b[C].y];
                const double2 cs_ = ln.csk(C, k + SH);
                bond_apply<LN::kCplx>(cs_.x, cs_.y, ln.sik(C, k + SH), a, d, ri[k], rj[k]);
            }
        }
    }
}

// U[k] <- B_l U[k] for the Hermitian Sym propagator B = C_L…C_2 (C_1 D C_1) C_2…C_L; the final
// colour's output is returned in (ri, rj) at the sites of the lane's bond in that colour.
template <int NCOL, int SH, class LN>
__device__ __forceinline__ void propagate_sym(double2 *U, int N, int nk, const LN &ln, double2 (&ri)[KMAX], double2 (&rj)[KMAX])
{
    if (NCOL == 1) {
        middle<SH, true>(U, N, nk, ln, ri, rj);
        return;
    }
    if (NCOL >= 4) stage<3 < NCOL ? 3 : 0, SH>(U, N, nk, ln);
    if (NCOL >= 3) stage<2 < NCOL ? 2 : 0, SH>(U, N, nk, ln);
    if (NCOL >= 2) stage<1 < NCOL ? 1 : 0, SH>(U, N, nk, ln);
    middle<SH, false>(U, N, nk, ln, ri, rj);
    if (NCOL >= 3) stage<1 < NCOL ? 1 : 0, SH>(U, N, nk, ln);
    if (NCOL >= 4) stage<2 < NCOL ? 2 : 0, SH>(U, N, nk, ln);
    last_stage<NCOL - 1, SH>(U, N, nk, ln, ri, rj);
}

template <int NCOL, int OP, bool CSV, bool CPLX = false>
__global__ void __launch_bounds__(1024) fdm_fast_kernel(FdmArgs a, FdmFast ff)
{
    extern __shared__ double2 U[];
    __shared__ double red[34];
    // XCD-aware order: consecutive tau-chunks of one system share halo slices and field lines, so
    // they are dealt to the same XCD (blocks b and b+8 share an XCD's L2)
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int chunk = bid % a.nchunk, sys = a.sys_first + bid / a.nchunk;
    stamp_begin(a.stamp);
    if (a.cg[sys].done) return;  // a.cg is never null (api_handle.hip, fdm_args / kpm_args: an all-zero state outside CG loops)
    const int w = sys / a.nrhs;
    const int Lt = a.Lt, N = a.N;
    const int l0 = chunk * a.Tc;
    const int nk = min(a.Tc, Lt - l0);
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double2 *csf = ff.csf + (size_t)w * Lt * ff.ptotal;
    constexpr bool FUSED = (OP == SMOQY_OP_MTM || OP == SMOQY_OP_MMT);
    const int K1 = FUSED ? nk + 1 : nk;                        // slices of the first propagate
    const int fbase = (OP == SMOQY_OP_MT) ? l0 + 1 : l0;       // field slice of register index 0
    const int ubase = (OP == SMOQY_OP_M || OP == SMOQY_OP_MTM) ? l0 - 1 : (OP == SMOQY_OP_MT ? l0 + 1 : l0);  // source slice of U[0]

    // ---- everything this workgroup needs from memory is requested here, up front ----
    static_assert(!CPLX || CSV, "complex hoppings are instantiated with one (cosh, sinh) set per slice only");
    LaneT<CSV, CPLX> ln;
    const double *csi = CPLX ? ff.csi + (size_t)w * Lt * ff.ptotal : nullptr;
    // hoppings that do not depend on τ (e.g. Holstein: t constant) are detected when the fields are
    // packed; the lane then fetches its (cosh, sinh) pair once instead of once per slice
    const bool cs_varies = CSV && ff.cs_varies[w] != 0;
#pragma unroll
    for (int c = 0; c < kFdmColours; ++c) {
        ln.on[c] = false;
        ln.b[c] = make_int2(0, 0);
        if (c < NCOL) {
            const int idx = ff.poff[c] + (int)threadIdx.x;
            if (idx < ff.poff[c + 1]) {
                ln.on[c] = true;
                ln.b[c] = ff.pbonds[idx];
#pragma unroll
                for (int k = 0; k < (CSV ? KMAX : 1); ++k)
                    if (k < K1 && (k == 0 || cs_varies)) ln.cs[c][k] = csf[(cs_varies ? (size_t)wrapl(fbase + k, Lt) : (size_t)0) * ff.ptotal + idx];  // τ-independent hoppings: every workgroup reads slice 0 (12 KB per walker, cache-resident) instead of its own copy
                if (CSV && !cs_varies) {
#pragma unroll
                    for (int k = 1; k < (CSV ? KMAX : 1); ++k) ln.cs[c][k] = ln.cs[c][0];
                }
                if constexpr (CPLX) {
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        ln.si[c][k] = 0.0;
                        if (k < K1) ln.si[c][k] = csi[(cs_varies ? (size_t)wrapl(fbase + k, Lt) : (size_t)0) * ff.ptotal + idx];
                    }
                }
            }
        }
    }
    const int2 s0 = ln.on[0] ? ff.psites[ff.poff[0] + (int)threadIdx.x] : make_int2(0, 0);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        ln.di[k] = ln.dj[k] = 1.0;
        if (k < K1 && ln.on[0]) {
            const double *e = expV + (size_t)wrapl(fbase + k, Lt) * N;
            ln.di[k] = e[s0.x];
            ln.dj[k] = e[s0.y];
        }
    }
    const int2 bL = ln.b[NCOL - 1];  // LDS positions of the lane's last-colour site pair
    const bool onL = ln.on[NCOL - 1];
    const int2 sL = onL ? ff.psites[ff.poff[NCOL - 1] + (int)threadIdx.x] : make_int2(0, 0);  // ... and the site ids
    // the "v" of v ∓ B v at the lane's own (last-colour) site pair
    // All but one of these slices are also among the slices staged into U (shifted by one), so only
    // that one is gathered from memory; the others are picked out of LDS after the fill.
    double2 vi[KMAX], vj[KMAX];
    const int vbase = (OP == SMOQY_OP_MMT) ? l0 - 1 : l0;  // slice of vi[0]
    constexpr int VSH = (OP == SMOQY_OP_M || OP == SMOQY_OP_MTM) ? 1 : -1;  // v[k] is U[k + VSH]
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        vi[k] = vj[k] = make_double2(0.0, 0.0);
        const int ku = k + VSH;
        if (k < K1 && onL && (ku < 0 || ku >= K1)) {
            const double2 *row = in + (size_t)wrapl(vbase + k, Lt) * sstride;
            vi[k] = row[sL.x];
            vj[k] = row[sL.y];
        }
    }
    for (int idx = threadIdx.x; idx < K1 * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        U[(size_t)k * N + ff.pos[i]] = in[(size_t)wrapl(ubase + k, Lt) * sstride + i];
    }
    __syncthreads();
    if (onL) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int ku = k + VSH;
            if (k < K1 && ku >= 0 && ku < K1) {
                vi[k] = U[(size_t)ku * N + bL.x];
                vj[k] = U[(size_t)ku * N + bL.y];
            }
        }
    }
    __syncthreads();  // the first stage overwrites U at other lanes' sites

    double2 ri[KMAX], rj[KMAX];
    double2 acc = make_double2(0.0, 0.0);
    propagate_sym<NCOL, 0>(U, N, K1, ln, ri, rj);
    if (!FUSED) {
        // M:  out[l] = v[l] ∓ B_l v[l-1]  (+ on the first slice);  Mᵀ: out[l] = v[l] ∓ B_{l+1} v[l+1]  (+ on the last)
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < nk && onL) {
                const int l = l0 + k;
                const bool wrap = (OP == SMOQY_OP_M) ? (l == 0) : (l == Lt - 1);
                const double2 oi = hopcomb(vi[k], ri[k], wrap, OP == SMOQY_OP_MT, a), oj = hopcomb(vj[k], rj[k], wrap, OP == SMOQY_OP_MT, a);
                double2 *row = out + (size_t)l * sstride;
                row[sL.x] = oi;
                acc.x += vi[k].x * oi.x + vi[k].y * oi.y;
                acc.y += vi[k].x * oi.y - vi[k].y * oi.x;
                if (bL.y != bL.x) {
                    row[sL.y] = oj;
                    acc.x += vj[k].x * oj.x + vj[k].y * oj.y;
                    acc.y += vj[k].x * oj.y - vj[k].y * oj.x;
                }
            }
        }
    } else {
        // y = first operator applied on nk+1 slices, kept in registers at the lane's own site pair
        double2 yi[KMAX], yj[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            yi[k] = yj[k] = make_double2(0.0, 0.0);
            if (k < K1) {
                const int l = wrapl(vbase + k, Lt);  // slice of y[k]
                const bool wrap = (OP == SMOQY_OP_MTM) ? (l == 0) : (l == Lt - 1);
                yi[k] = hopcomb(vi[k], ri[k], wrap, OP == SMOQY_OP_MMT, a);
                yj[k] = hopcomb(vj[k], rj[k], wrap, OP == SMOQY_OP_MMT, a);
            }
        }
        // hand over to the second operator: MᵀM propagates y[1..nk] with fields of slices l0+1..,
        // MMᵀ propagates y[0..nk-1] with fields of slices l0..
        __syncthreads();  // every lane is done reading U in the last stage
        if (onL) {
#pragma unroll
            for (int k = 0; k < KMAX - 1; ++k) {
                if (k < nk) {
                    double2 *row = U + (size_t)k * N;
                    row[bL.x] = (OP == SMOQY_OP_MTM) ? yi[k + 1] : yi[k];
                    row[bL.y] = (OP == SMOQY_OP_MTM) ? yj[k + 1] : yj[k];
                }
            }
        }
        __syncthreads();
        if (OP == SMOQY_OP_MTM) propagate_sym<NCOL, 1>(U, N, nk, ln, ri, rj);
        else propagate_sym<NCOL, 0>(U, N, nk, ln, ri, rj);
        // the dot(in, out) partial needs `in` on the output slices at the own site pair: for MᵀM
        // that is vi/vj (slices l0+k); for MMᵀ vi[k+1]
#pragma unroll
        for (int k = 0; k < KMAX - 1; ++k) {
            if (k < nk && onL) {
                const int l = l0 + k;
                const bool wrap = (OP == SMOQY_OP_MTM) ? (l == Lt - 1) : (l == 0);
                const double2 bi = (OP == SMOQY_OP_MTM) ? yi[k] : yi[k + 1], bj = (OP == SMOQY_OP_MTM) ? yj[k] : yj[k + 1];
                const double2 oi = hopcomb(bi, ri[k], wrap, OP == SMOQY_OP_MTM, a), oj = hopcomb(bj, rj[k], wrap, OP == SMOQY_OP_MTM, a);
                const double2 pi = (OP == SMOQY_OP_MTM) ? vi[k] : vi[k + 1], pj = (OP == SMOQY_OP_MTM) ? vj[k] : vj[k + 1];
                double2 *row = out + (size_t)l * sstride;
                row[sL.x] = oi;
                acc.x += pi.x * oi.x + pi.y * oi.y;
                acc.y += pi.x * oi.y - pi.y * oi.x;
                if (bL.y != bL.x) {
                    row[sL.y] = oj;
                    acc.x += pj.x * oj.x + pj.y * oj.y;
                    acc.y += pj.x * oj.y - pj.y * oj.x;
                }
            }
        }
    }
    if (a.partial) {
        // wavefront shuffles, then one LDS hop across the workgroup's wavefronts
        for (int off = 32; off > 0; off >>= 1) {
            acc.x += __shfl_down(acc.x, off, 64);
            acc.y += __shfl_down(acc.y, off, 64);
        }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
        if (lane == 0) { red[2 * wave] = acc.x; red[2 * wave + 1] = acc.y; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double2 t = make_double2(0.0, 0.0);
            for (int q = 0; q < nwave; ++q) { t.x += red[2 * q]; t.y += red[2 * q + 1]; }
            a.partial[(size_t)sys * a.nchunk + chunk] = t;
        }
    }
    stamp_end(a.stamp);
}

// ---------------------------------------------------------------------------------------------
// Asym form (round 2): B_l = D_l Γ_l, B_lᴴ = Γ_lᴴ D_l with Γ = C_{L-1} … C_1 C_0 (colour 0 first), src/FermionDetMatrix.jl:430-466, 528-563.
// Same machinery as the Sym kernel — padded per-colour bond lists, the (cosh, sinh) pairs of a lane's bonds in registers, slices in LDS,
// the last stage of a propagate left in registers — with two differences: there is no folded middle stage (L stages per B instead of 2L-1),
// and the two operators end on DIFFERENT colours: M leaves its result with the owner of the last colour's bond (where D is applied), Mᴴ
// with the owner of the first colour's bond.  The fused products therefore hand y = Mv (or Mᴴv) over through a second LDS image Y, which the
// other owner reads for its `y ∓ B y` combine; the first stage of the second operator runs on the registers that still hold y.
// ---------------------------------------------------------------------------------------------
template <int NCOL, int OP>
__global__ void __launch_bounds__(1024) fdm_fast_asym_kernel(FdmArgs a, FdmFast ff)
{
    extern __shared__ double2 U[];
    __shared__ double red[34];
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int chunk = bid % a.nchunk, sys = a.sys_first + bid / a.nchunk;
    stamp_begin(a.stamp);
    if (a.cg[sys].done) return;  // a.cg is never null (api_handle.hip, fdm_args / kpm_args: an all-zero state outside CG loops)
    const int w = sys / a.nrhs;
    const int Lt = a.Lt, N = a.N;
    const int l0 = chunk * a.Tc;
    const int nk = min(a.Tc, Lt - l0);
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double2 *csf = ff.csf + (size_t)w * Lt * ff.ptotal;
    constexpr bool FUSED = (OP == SMOQY_OP_MTM || OP == SMOQY_OP_MMT);
    constexpr bool M_FIRST = (OP == SMOQY_OP_M || OP == SMOQY_OP_MTM);  // the first (or only) operator is M: forward colour order
    constexpr int CL = NCOL - 1;
    double2 *Y = U + (size_t)(a.Tc + 1) * N;                   // fused products only
    const int K1 = FUSED ? nk + 1 : nk;
    const int fbase = (OP == SMOQY_OP_MT) ? l0 + 1 : l0;       // field slice of register index 0
    const int ubase = M_FIRST ? l0 - 1 : (OP == SMOQY_OP_MT ? l0 + 1 : l0);  // source slice of U[0]

    Lane ln;
    const bool cs_varies = ff.cs_varies[w] != 0;
#pragma unroll
    for (int c = 0; c < kFdmColours; ++c) {
        ln.on[c] = false;
        ln.b[c] = make_int2(0, 0);
        if (c < NCOL) {
            const int idx = ff.poff[c] + (int)threadIdx.x;
            if (idx < ff.poff[c + 1]) {
                ln.on[c] = true;
                ln.b[c] = ff.pbonds[idx];
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K1 && (k == 0 || cs_varies)) ln.cs[c][k] = csf[(cs_varies ? (size_t)wrapl(fbase + k, Lt) : (size_t)0) * ff.ptotal + idx];  // τ-independent hoppings: every workgroup reads slice 0 (12 KB per walker, cache-resident) instead of its own copy
                if (!cs_varies) {
#pragma unroll
                    for (int k = 1; k < KMAX; ++k) ln.cs[c][k] = ln.cs[c][0];
                }
            }
        }
    }
    const bool on0 = ln.on[0], onL = ln.on[CL];
    const int2 b0 = ln.b[0], bL = ln.b[CL];
    const int2 s0 = on0 ? ff.psites[ff.poff[0] + (int)threadIdx.x] : make_int2(0, 0);
    const int2 sL = onL ? ff.psites[ff.poff[CL] + (int)threadIdx.x] : make_int2(0, 0);
    // exp(-ΔτV) at the sites of the lane's LAST-colour bond: D sits next to that colour in both B = DΓ and Bᴴ = ΓᴴD
    double dLi[KMAX], dLj[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        dLi[k] = dLj[k] = 1.0;
        if (k < K1 && onL) {
            const double *e = expV + (size_t)wrapl(fbase + k, Lt) * N;
            dLi[k] = e[sL.x];
            dLj[k] = e[sL.y];
        }
    }
    // phase-1 owner: last colour's bond for M, first colour's bond for Mᴴ
    const bool onA = M_FIRST ? onL : on0;
    const int2 bA = M_FIRST ? bL : b0, sA = M_FIRST ? sL : s0;
    double2 vi[KMAX], vj[KMAX];
    const int vbase = (OP == SMOQY_OP_MMT) ? l0 - 1 : l0;
    constexpr int VSH = M_FIRST ? 1 : -1;  // v[k] is U[k + VSH]
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        vi[k] = vj[k] = make_double2(0.0, 0.0);
        const int ku = k + VSH;
        if (k < K1 && onA && (ku < 0 || ku >= K1)) {
            const double2 *row = in + (size_t)wrapl(vbase + k, Lt) * sstride;
            vi[k] = row[sA.x];
            vj[k] = row[sA.y];
        }
    }
    for (int idx = threadIdx.x; idx < K1 * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        U[(size_t)k * N + ff.pos[i]] = in[(size_t)wrapl(ubase + k, Lt) * sstride + i];
    }
    __syncthreads();
    if (onA) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int ku = k + VSH;
            if (k < K1 && ku >= 0 && ku < K1) {
                vi[k] = U[(size_t)ku * N + bA.x];
                vj[k] = U[(size_t)ku * N + bA.y];
            }
        }
    }
    __syncthreads();

    double2 ri[KMAX], rj[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) ri[k] = rj[k] = make_double2(0.0, 0.0);
    // ---- first operator on K1 slices --------------------------------------------------------------------
    if (M_FIRST) {  // B = D Γ: colours 0 … L-2 through LDS, the last colour into registers, then D
        if (NCOL >= 2) stage<0, 0>(U, N, K1, ln);
        if (NCOL >= 3) stage<1 < NCOL ? 1 : 0, 0>(U, N, K1, ln);
        if (NCOL >= 4) stage<2 < NCOL ? 2 : 0, 0>(U, N, K1, ln);
        last_stage<CL, 0>(U, N, K1, ln, ri, rj);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { ri[k] = scl(dLi[k], ri[k]); rj[k] = scl(dLj[k], rj[k]); }
    } else {        // Bᴴ = Γᴴ D: D and the last colour on the owner's pair, colours L-2 … 1 through LDS, colour 0 into registers
        if (onL) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K1) {
                    double2 *row = U + (size_t)k * N;
                    const double2 x = scl(dLi[k], row[bL.x]), y = scl(dLj[k], row[bL.y]);
                    const double c = ln.cs[CL][k].x, s = ln.cs[CL][k].y;
                    if (NCOL == 1) { ri[k] = lin(c, x, s, y); rj[k] = lin(c, y, s, x); }
                    else { row[bL.x] = lin(c, x, s, y); row[bL.y] = lin(c, y, s, x); }
                }
            }
        }
        if (NCOL >= 2) {
            __syncthreads();
            if (NCOL >= 4) stage<2 < NCOL ? 2 : 0, 0>(U, N, K1, ln);
            if (NCOL >= 3) stage<1 < NCOL ? 1 : 0, 0>(U, N, K1, ln);
            last_stage<0, 0>(U, N, K1, ln, ri, rj);
        }
    }
    double2 acc = make_double2(0.0, 0.0);
    if (!FUSED) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < nk && onA) {
                const int l = l0 + k;
                const bool wrap = (OP == SMOQY_OP_M) ? (l == 0) : (l == Lt - 1);
                const double2 oi = hopcomb(vi[k], ri[k], wrap, OP == SMOQY_OP_MT, a), oj = hopcomb(vj[k], rj[k], wrap, OP == SMOQY_OP_MT, a);
                double2 *row = out + (size_t)l * sstride;
                row[sA.x] = oi;
                acc.x += vi[k].x * oi.x + vi[k].y * oi.y;
                acc.y += vi[k].x * oi.y - vi[k].y * oi.x;
                if (bA.y != bA.x) {
                    row[sA.y] = oj;
                    acc.x += vj[k].x * oj.x + vj[k].y * oj.y;
                    acc.y += vj[k].x * oj.y - vj[k].y * oj.x;
                }
            }
        }
    } else {
        // y = first operator on K1 slices, in the registers of the phase-1 owner
        double2 yi[KMAX], yj[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            yi[k] = yj[k] = make_double2(0.0, 0.0);
            if (k < K1) {
                const int l = wrapl(vbase + k, Lt);
                const bool wrap = M_FIRST ? (l == 0) : (l == Lt - 1);
                yi[k] = hopcomb(vi[k], ri[k], wrap, !M_FIRST, a);
                yj[k] = hopcomb(vj[k], rj[k], wrap, !M_FIRST, a);
            }
        }
        __syncthreads();  // every lane is done reading U
        // phase-2 owner: the other end of the colour chain
        const bool onB = M_FIRST ? on0 : onL;
        const int2 bB = M_FIRST ? b0 : bL, sB = M_FIRST ? s0 : sL;
        if (M_FIRST) {
            // MᴴM: out[l0+k] = y[k] − conj(ph) Bᴴ_{l0+k+1} y[k+1]; D and the last colour act on the registers that hold y[k+1]
            if (onL) {
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k) {
                    if (k < nk) {
                        const double2 x = scl(dLi[k + 1], yi[k + 1]), y = scl(dLj[k + 1], yj[k + 1]);
                        const double c = ln.cs[CL][k + 1].x, s = ln.cs[CL][k + 1].y;
                        const double2 nx = lin(c, x, s, y), ny = lin(c, y, s, x);
                        if (NCOL == 1) { ri[k] = nx; rj[k] = ny; }
                        else { U[(size_t)k * N + bL.x] = nx; U[(size_t)k * N + bL.y] = ny; }
                        Y[(size_t)k * N + bL.x] = yi[k];
                        Y[(size_t)k * N + bL.y] = yj[k];
                    }
                }
            }
            __syncthreads();
            if (NCOL >= 2) {
                if (NCOL >= 4) stage<2 < NCOL ? 2 : 0, 1>(U, N, nk, ln);
                if (NCOL >= 3) stage<1 < NCOL ? 1 : 0, 1>(U, N, nk, ln);
                last_stage<0, 1>(U, N, nk, ln, ri, rj);
            }
        } else {
            // MMᴴ: out[l0+k] = y[k+1] − ph B_{l0+k} y[k]; the first colour acts on the registers that hold y[k]
            if (on0) {
#pragma unroll
                for (int k = 0; k < KMAX - 1; ++k) {
                    if (k < nk) {
                        const double c = ln.cs[0][k].x, s = ln.cs[0][k].y;
                        const double2 nx = lin(c, yi[k], s, yj[k]), ny = lin(c, yj[k], s, yi[k]);
                        if (NCOL == 1) { ri[k] = scl(dLi[k], nx); rj[k] = scl(dLj[k], ny); }
                        else { U[(size_t)k * N + b0.x] = nx; U[(size_t)k * N + b0.y] = ny; }
                        Y[(size_t)k * N + b0.x] = yi[k + 1];
                        Y[(size_t)k * N + b0.y] = yj[k + 1];
                    }
                }
            }
            __syncthreads();
            if (NCOL >= 2) {
                if (NCOL >= 3) stage<1 < NCOL ? 1 : 0, 0>(U, N, nk, ln);
                if (NCOL >= 4) stage<2 < NCOL ? 2 : 0, 0>(U, N, nk, ln);
                last_stage<CL, 0>(U, N, nk, ln, ri, rj);
#pragma unroll
                for (int k = 0; k < KMAX; ++k) { ri[k] = scl(dLi[k], ri[k]); rj[k] = scl(dLj[k], rj[k]); }
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX - 1; ++k) {
            if (k < nk && onB) {
                const int l = l0 + k;
                const bool wrap = M_FIRST ? (l == Lt - 1) : (l == 0);
                const double2 bi = Y[(size_t)k * N + bB.x], bj = Y[(size_t)k * N + bB.y];
                const double2 oi = hopcomb(bi, ri[k], wrap, M_FIRST, a), oj = hopcomb(bj, rj[k], wrap, M_FIRST, a);
                const double2 *prow = in + (size_t)l * sstride;  // dot(in, out) needs `in` at the phase-2 owner's sites (L2 hits)
                const double2 pi = prow[sB.x], pj = prow[sB.y];
                double2 *row = out + (size_t)l * sstride;
                row[sB.x] = oi;
                acc.x += pi.x * oi.x + pi.y * oi.y;
                acc.y += pi.x * oi.y - pi.y * oi.x;
                if (bB.y != bB.x) {
                    row[sB.y] = oj;
                    acc.x += pj.x * oj.x + pj.y * oj.y;
                    acc.y += pj.x * oj.y - pj.y * oj.x;
                }
            }
        }
    }
    if (a.partial) {
        for (int off = 32; off > 0; off >>= 1) {
            acc.x += __shfl_down(acc.x, off, 64);
            acc.y += __shfl_down(acc.y, off, 64);
        }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
        if (lane == 0) { red[2 * wave] = acc.x; red[2 * wave + 1] = acc.y; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double2 t = make_double2(0.0, 0.0);
            for (int q = 0; q < nwave; ++q) { t.x += red[2 * q]; t.y += red[2 * q + 1]; }
            a.partial[(size_t)sys * a.nchunk + chunk] = t;
        }
    }
    stamp_end(a.stamp);
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the fused MᵀM (round 3; src/FermionDetMatrix.jl:329-340 = mul_Mt!(mul_M!)), for launches whose working set is
// beyond the caches.  fdm_fast_kernel gives every τ-chunk its own workgroup: load four slices, wait, run twenty-odd barrier stages,
// store two slices, exit — at 128 systems per launch the loads of a workgroup are not in flight while it computes, the launch is as
// long as (workgroup lifetime) x (workgroups per slot) and reaches 0.46 of the measured copy rate.  Here a workgroup WALKS a run of
// consecutive slices of one system and the two propagates of the fused product are software-pipelined against each other:
//
//     y[m]     = v[m] − h·B_m v[m−1]                     (row m of M;  P1(m))
//     out[m−1] = y[m−1] − h̄·B_m y[m]                     (row m−1 of Mᵀ; P2(m))        — both use the fields of slice m only
//
// Iteration j applies B_{j+1} to v[j] (→ y[j+1]) and B_j to y[j] (→ out[j−1]) in the SAME barrier stages (two independent slices per
// stage), while the slices v[j+2], v[j+3] are on their way from memory into registers (loaded two iterations ahead, 16 bytes per lane
// each) and land in a free LDS image at the top of a later iteration.  Every input slice is read ONCE per run (+2 halo slices per run,
// where the chunked kernel re-reads two of four), the memory stream of a workgroup never stops, and an output slice costs five
// barriers (chunked: 6.5).  Four LDS images of one slice each (32 KB at N = 512): slot-1 input v[j], slot-2 input y[j], the landed
// v[j+1] (its values at the lane's own site pair feed the `v − B v` combine), and the image that receives y[j+1] for the next iteration.
// p·Ap is accumulated as Σ|y[m]|² over the run's own slices — (Mp)·(Mp) — which is the same number as p·(MᵀM p) up to rounding, real and
// non-negative by construction, and needs no `p` at the own sites two iterations later.
// Arithmetic per site is that of fdm_fast_kernel (same stage order, same folded C₁DC₁): outputs are bit-identical.
// ---------------------------------------------------------------------------------------------
// FULL: every colour is a perfect matching of the lattice (padded lists of exactly blockDim.x two-site bonds, N = 2·blockDim.x — the host
// checks it): no lane is ever switched off, so every LDS access and every store of the pipeline is unconditional.  Besides the saved
// exec-mask juggling this keeps the compiler's count of outstanding memory operations exact: with a store inside a branch it falls back to
// waiting for all but the newest three, i.e. for the slice requested one iteration ago.
// LB: the launch bound.  1024 lanes cap the kernel at 128 VGPRs, which two field sets of three or four colours (CSV) do not fit beside
// the pipeline; lattices of up to 256 padded bonds per colour (every BASELINE.json lattice) are launched with the 256-lane instantiation,
// which may use what it needs.
template <int NCOL, bool CSV, bool FULL, int LB>
__global__ void __launch_bounds__(LB) fdm_stream_kernel(FdmArgs a, FdmFast ff)
{
    extern __shared__ double2 U[];
    __shared__ double red[34];
    constexpr int CL = NCOL - 1;
    const int Lt = a.Lt, N = a.N, Tn = blockDim.x, t = threadIdx.x;
    const int R = a.run_len, nrun = (Lt + R - 1) / R;
    // XCD-aware order (as fdm_fast_kernel): consecutive runs of one system share their halo slices
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int run = bid % nrun, sys = a.sys_first + bid / nrun;
    stamp_begin(a.stamp);
    const int sys_done = a.cg[sys].done;  // (a.cg is never null)  // acted on in the prologue, with the bond program and the first two slices already in flight
    const int w = sys / a.nrhs;
    const int la = run * R, lb = min(Lt, la + R);  // output slices [la, lb)
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double2 *csf = ff.csf + (size_t)w * Lt * ff.ptotal;
    const bool cs_varies = CSV && ff.cs_varies[w] != 0;

    // ---- the lane's bond program ----
    int2 b[NCOL];
    bool on[NCOL];
    int cidx[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        cidx[c] = ff.poff[c] + t;
        on[c] = FULL || cidx[c] < ff.poff[c + 1];
        if (!on[c]) cidx[c] = ff.poff[c];
        b[c] = ff.pbonds[cidx[c]];
    }
    const int2 s0 = ff.psites[cidx[0]];
    const int2 bL = b[CL];
    const bool onL = FULL || on[CL];
    const int2 sL = ff.psites[cidx[CL]];
    const bool twoL = FULL || bL.y != bL.x;  // the lane's last-colour bond has two distinct sites
    // a slice in load layout: lane t holds elements t and t + Tn (N <= 2 Tn: every colour's padded list covers all sites)
    const int e0 = t, e1 = t + Tn;
    const bool ok0 = FULL || e0 < N, ok1 = FULL || e1 < N;
    const int p0 = ok0 ? ff.pos[e0] : 0, p1 = ok1 ? ff.pos[e1] : 0;  // their LDS positions
#define X_(i_) (U + (size_t)(i_) * (size_t)N)  // image i (computed, not looked up: a dynamically indexed pointer array would live in scratch memory)

    // fields of one slice at the lane's bonds: (cosh, sinh) per colour and exp(-ΔτV) at the colour-0 pair.  Three sets: slot 1, slot 2 and
    // the slice after them (loaded one iteration ahead).  τ-independent hoppings (CSV = false, or a walker whose table says so): one pair
    // per colour for the whole run, read from slice 0 of the packed table.
    constexpr int NC = CSV ? NCOL : 1;
    double2 cs_const[NCOL], cs1[NC], cs2[NC], csn[NC];
    double d1i, d1j, d2i, d2j, dni, dnj;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) cs_const[c] = csf[cidx[c]];  // used by CSV = false only (the host has shown the hoppings to be τ-independent)
    (void)cs_varies;
    // Every global load of the pipeline is UNCONDITIONAL (clamped addresses, values of switched-off lanes never used): a load inside a
    // branch makes the compiler lose count of what is in flight and fall back to `s_waitcnt vmcnt(0)` — a full memory round trip per
    // iteration, which is what the first form of this kernel measured (3.9 µs per iteration).
#define LOAD_FLD(cs_, di_, dj_, m_)                                                                                   \
    {                                                                                                                 \
        const int l_ = wrapl((m_), Lt);                                                                               \
        if (CSV) {                                                                                                    \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) cs_[c] = csf[(size_t)l_ * ff.ptotal + cidx[c]];           \
        }                                                                                                             \
        if (a.nt_fields) {                                                                                            \
            di_ = __builtin_nontemporal_load(&expV[(size_t)l_ * N + s0.x]);                                           \
            dj_ = __builtin_nontemporal_load(&expV[(size_t)l_ * N + s0.y]);                                           \
        } else {                                                                                                      \
            di_ = expV[(size_t)l_ * N + s0.x];                                                                        \
            dj_ = expV[(size_t)l_ * N + s0.y];                                                                        \
        }                                                                                                             \
    }
#define ROTATE_FLD()                                                                                                  \
    {                                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < NC; ++c) { cs2[c] = cs1[c]; cs1[c] = csn[c]; }                          \
        d2i = d1i; d2j = d1j; d1i = dni; d1j = dnj;                                                                   \
    }
    const int e0c = min(e0, N - 1), e1c = min(e1, N - 1);
    // (macros on named scalars, not lambdas on arrays: the prefetch registers must not be demoted to scratch memory)
#define LOAD_SLICE(r0_, r1_, m_)                                                  \
    {                                                                             \
        const double2 *row_ = in + (size_t)wrapl((m_), Lt) * sstride;             \
        r0_ = row_[e0c];                                                          \
        r1_ = row_[e1c];                                                          \
    }
#define LAND(r0_, r1_, img_)                                                      \
    {                                                                             \
        double2 *img__ = (img_);                                                  \
        if (ok0) img__[p0] = r0_;                                                 \
        if (ok1) img__[p1] = r1_;                                                 \
    }
    // the stage chain of B = C_L…C_2 (C_1 D C_1) C_2…C_L on two independent slices at once: slot 1 = image A with fields f1, slot 2 =
    // image Bm with fields f2 (either may be switched off); the last colour's results come back in registers at the lane's own pair
#define CS1(c_) (CSV ? cs1[CSV ? (c_) : 0] : cs_const[c_])
#define CS2(c_) (CSV ? cs2[CSV ? (c_) : 0] : cs_const[c_])
#define ST_PLAIN(c_)                                                                                   \
    {                                                                                                  \
        if (FULL || on[c_]) {                                                                                  \
            if (e1on) {                                                                                \
                const double2 x_ = A[b[c_].x], y_ = A[b[c_].y], k_ = CS1(c_);                          \
                A[b[c_].x] = lin(k_.x, x_, k_.y, y_);                                                  \
                A[b[c_].y] = lin(k_.x, y_, k_.y, x_);                                                  \
            }                                                                                          \
            if (e2on) {                                                                                \
                const double2 x_ = Bm[b[c_].x], y_ = Bm[b[c_].y], k_ = CS2(c_);                        \
                Bm[b[c_].x] = lin(k_.x, x_, k_.y, y_);                                                 \
                Bm[b[c_].y] = lin(k_.x, y_, k_.y, x_);                                                 \
            }                                                                                          \
        }                                                                                              \
        __syncthreads();                                                                               \
    }
#define ST_MIDDLE(last_)                                                                               \
    {                                                                                                  \
        if (FULL || on[0]) {                                                                                   \
            if (e1on) {                                                                                \
                const double2 x_ = A[b[0].x], y_ = A[b[0].y], k_ = CS1(0);                             \
                const double2 u_ = scl(d1i, lin(k_.x, x_, k_.y, y_)), v_ = scl(d1j, lin(k_.x, y_, k_.y, x_)); \
                if (last_) { r1i = lin(k_.x, u_, k_.y, v_); r1j = lin(k_.x, v_, k_.y, u_); }           \
                else { A[b[0].x] = lin(k_.x, u_, k_.y, v_); A[b[0].y] = lin(k_.x, v_, k_.y, u_); }     \
            }                                                                                          \
            if (e2on) {                                                                                \
                const double2 x_ = Bm[b[0].x], y_ = Bm[b[0].y], k_ = CS2(0);                           \
                const double2 u_ = scl(d2i, lin(k_.x, x_, k_.y, y_)), v_ = scl(d2j, lin(k_.x, y_, k_.y, x_)); \
                if (last_) { r2i = lin(k_.x, u_, k_.y, v_); r2j = lin(k_.x, v_, k_.y, u_); }           \
                else { Bm[b[0].x] = lin(k_.x, u_, k_.y, v_); Bm[b[0].y] = lin(k_.x, v_, k_.y, u_); }   \
            }                                                                                          \
        }                                                                                              \
        if (!(last_)) __syncthreads();                                                                 \
    }
#define ST_LAST(c_)                                                                                    \
    {                                                                                                  \
        if (FULL || on[c_]) {                                                                                  \
            if (e1on) {                                                                                \
                const double2 x_ = A[b[c_].x], y_ = A[b[c_].y], k_ = CS1(c_);                          \
                r1i = lin(k_.x, x_, k_.y, y_);                                                         \
                r1j = lin(k_.x, y_, k_.y, x_);                                                         \
            }                                                                                          \
            if (e2on) {                                                                                \
                const double2 x_ = Bm[b[c_].x], y_ = Bm[b[c_].y], k_ = CS2(c_);                        \
                r2i = lin(k_.x, x_, k_.y, y_);                                                         \
                r2j = lin(k_.x, y_, k_.y, x_);                                                         \
            }                                                                                          \
        }                                                                                              \
    }
#define PROPAGATE2()                                                                                   \
    {                                                                                                  \
        if (NCOL == 1) {                                                                               \
            ST_MIDDLE(true)                                                                            \
        } else {                                                                                       \
            if (NCOL >= 4) ST_PLAIN((3 < NCOL ? 3 : 0))                                                \
            if (NCOL >= 3) ST_PLAIN((2 < NCOL ? 2 : 0))                                                \
            if (NCOL >= 2) ST_PLAIN((1 < NCOL ? 1 : 0))                                                \
            ST_MIDDLE(false)                                                                           \
            if (NCOL >= 3) ST_PLAIN((1 < NCOL ? 1 : 0))                                                \
            if (NCOL >= 4) ST_PLAIN((2 < NCOL ? 2 : 0))                                                \
            ST_LAST(CL)                                                                                \
        }                                                                                              \
    }

    // ---- prologue: slices la-1, la, la+1 straight into their images, the next two on their way; P1(la+1) and P1(la) together ----
    // Order of issue: what the steady-state loop will wait for first (the two prefetched slices, the fields of slice la+2) goes out FIRST,
    // the slices and fields the prologue itself needs behind it.  Results return in issue order, so by the time the prologue has its
    // own data the prefetches have landed as well, and the loop is entered with nothing of them outstanding — otherwise the compiler
    // merges "entered from the prologue" with "came around the loop" and settles for the shorter of the two distances at every wait.
    double2 pfa0, pfa1, pfb0, pfb1;
    LOAD_SLICE(pfa0, pfa1, min(la + 2, lb))  // needed as slices up to lb (the last one only for its values at the own sites)
    LOAD_SLICE(pfb0, pfb1, min(la + 3, lb))
    asm volatile("" ::: "memory");  // compiler fence: keeps the loads above on this side of the early return (no instruction, no wait)
    if (sys_done) return;  // workgroup-uniform; nothing has been stored yet
    LOAD_FLD(csn, dni, dnj, la + 2)
    LOAD_FLD(cs2, d2i, d2j, la)        // slot 2 of the prologue: B_la on v[la-1]
    LOAD_FLD(cs1, d1i, d1j, la + 1)    // slot 1: B_{la+1} on v[la]
    {
        double2 t0, t1, t2, t3, t4, t5;
        LOAD_SLICE(t0, t1, la - 1)
        LOAD_SLICE(t2, t3, la)
        LOAD_SLICE(t4, t5, la + 1)
        LAND(t0, t1, X_(0))
        LAND(t2, t3, X_(1))
        LAND(t4, t5, X_(2))
    }
    __syncthreads();
    double2 vown_i = make_double2(0.0, 0.0), vown_j = vown_i, vnext_i = vown_i, vnext_j = vown_i;
    if (onL) { vown_i = X_(1)[bL.x]; vown_j = X_(1)[bL.y]; vnext_i = X_(2)[bL.x]; vnext_j = X_(2)[bL.y]; }  // v[la], v[la+1] at the own pair (the first stage
                                                                                                       // writes exactly these positions of X_(1): same lane, no race)
    double2 r1i = make_double2(0.0, 0.0), r1j = r1i, r2i = r1i, r2j = r1i;
    double accr = 0.0;
    double2 yprev_i, yprev_j, ycur_i, ycur_j;
    {
        double2 *A = X_(1), *Bm = X_(0);
        const bool e1on = true, e2on = true;
        PROPAGATE2()
        yprev_i = hopcomb(vown_i, r2i, wrapl(la, Lt) == 0, false, a);       // y[la]
        yprev_j = hopcomb(vown_j, r2j, wrapl(la, Lt) == 0, false, a);
        ycur_i = hopcomb(vnext_i, r1i, wrapl(la + 1, Lt) == 0, false, a);   // y[la+1]
        ycur_j = hopcomb(vnext_j, r1j, wrapl(la + 1, Lt) == 0, false, a);
        if (onL) {
            accr += yprev_i.x * yprev_i.x + yprev_i.y * yprev_i.y;
            if (twoL) accr += yprev_j.x * yprev_j.x + yprev_j.y * yprev_j.y;
            X_(3)[bL.x] = ycur_i;  // hand y[la+1] over to slot 2 of the next iteration (X_(3) is free)
            X_(3)[bL.y] = ycur_j;
        }
        __syncthreads();
    }
    int iV = 2, iY = 3, iN = 0, iF = 1;  // images: v[j] (slot 1), y[j] (slot 2), landing of v[j+1], receiver of y[j+1]
    ROTATE_FLD()
    // ---- steady state: iteration j does P1(j+1) (unless j = lb) and P2(j).  Written out twice per trip so that the two prefetch register
    // sets alternate without a copy (a copy would be a use, i.e. a wait for the younger load) ----
#define STREAM_ITER(pf0_, pf1_, j_)                                                                                       \
    {                                                                                                                 \
        const int jj = (j_);                                                                                          \
        const bool e1on = jj < lb, e2on = true;                                                                       \
        /* issue order = age at first use: fields of slice jj+2 (next iteration) first, the slice v[jj+3] behind them */ \
        LOAD_FLD(csn, dni, dnj, min(jj + 2, lb))                                                                      \
        LAND(pf0_, pf1_, X_(iN)) /* v[jj+1], loaded two iterations ago (unused in the last iteration: the image is free) */ \
        LOAD_SLICE(pf0_, pf1_, min(jj + 3, lb))                                                                       \
        double2 *A = X_(iV), *Bm = X_(iY);                                                                            \
        PROPAGATE2()                                                                                                  \
        if (onL) { /* out[jj-1] = y[jj-1] − h̄ B_jj y[jj] */                                                            \
            const bool wrap = (jj - 1) == Lt - 1;                                                                     \
            const double2 oi = hopcomb(yprev_i, r2i, wrap, true, a), oj = hopcomb(yprev_j, r2j, wrap, true, a);       \
            double2 *row = out + (size_t)(jj - 1) * sstride;                                                          \
            row[sL.x] = oi;                                                                                           \
            if (twoL) row[sL.y] = oj;                                                                         \
        }                                                                                                             \
        yprev_i = ycur_i; yprev_j = ycur_j;                                                                           \
        if (e1on) {                                                                                                   \
            /* y[jj+1] = v[jj+1] − h B_{jj+1} v[jj]; v[jj+1] at the own pair comes out of the image it landed in */     \
            double2 vi_ = make_double2(0.0, 0.0), vj_ = vi_;                                                          \
            if (onL) { vi_ = X_(iN)[bL.x]; vj_ = X_(iN)[bL.y]; }                                                      \
            ycur_i = hopcomb(vi_, r1i, wrapl(jj + 1, Lt) == 0, false, a);                                             \
            ycur_j = hopcomb(vj_, r1j, wrapl(jj + 1, Lt) == 0, false, a);                                             \
            if (onL) {                                                                                                \
                accr += yprev_i.x * yprev_i.x + yprev_i.y * yprev_i.y; /* |y[jj]|², jj < lb: a slice of this run */    \
                if (twoL) accr += yprev_j.x * yprev_j.x + yprev_j.y * yprev_j.y;                              \
                X_(iF)[bL.x] = ycur_i;                                                                                \
                X_(iF)[bL.y] = ycur_j;                                                                                \
            }                                                                                                         \
            __syncthreads(); /* y[jj+1] visible; every lane is done with the images of this iteration */              \
            const int oV = iV, oY = iY;                                                                               \
            iV = iN; iY = iF; iN = oV; iF = oY;                                                                       \
            ROTATE_FLD()                                                                                              \
        }                                                                                                             \
    }
    // (both halves in every trip, the odd one out behind the loop: a conditional second half would give the compiler a path around the loop
    // on which the first set's load is only one iteration old, and it would wait for that distance everywhere)
    int j = la + 1;
    for (; j + 1 <= lb; j += 2) {
        STREAM_ITER(pfa0, pfa1, j)
        STREAM_ITER(pfb0, pfb1, j + 1)
    }
    if (j <= lb) STREAM_ITER(pfa0, pfa1, j)
#undef STREAM_ITER
#undef LAND
#undef LOAD_SLICE
#undef X_
#undef PROPAGATE2
#undef ST_LAST
#undef ST_MIDDLE
#undef ST_PLAIN
#undef CS1
#undef CS2
#undef ROTATE_FLD
#undef LOAD_FLD
    if (a.partial) {
        for (int off = 32; off > 0; off >>= 1) accr += __shfl_down(accr, off, 64);
        const int wave = t >> 6, lane = t & 63, nwave = (Tn + 63) >> 6;
        if (lane == 0) red[wave] = accr;
        __syncthreads();
        if (t == 0) {
            double s = 0.0;
            for (int q = 0; q < nwave; ++q) s += red[q];
            // the consumers reduce a.nchunk partials per system: the run's sum goes to its first chunk, its other chunks are zero
            const int c0 = la / a.Tc, c1 = (lb + a.Tc - 1) / a.Tc;
            a.partial[(size_t)sys * a.nchunk + c0] = make_double2(s, 0.0);
            for (int c = c0 + 1; c < c1; ++c) a.partial[(size_t)sys * a.nchunk + c] = make_double2(0.0, 0.0);
        }
    }
    stamp_end(a.stamp);
}

template <int NCOL, bool CSV>
void launch_stream_ncol(hipStream_t st, const FdmArgs &a, const FdmFast &ff)
{
    const int nrun = (a.Lt + a.run_len - 1) / a.run_len;
    const dim3 grid((unsigned)(nrun * a.sys_count)), block((unsigned)ff.threads);
    const size_t lds = sizeof(double2) * 4 * (size_t)a.N;
    if (ff.threads <= 256) {
        if (ff.full) hipLaunchKernelGGL((fdm_stream_kernel<NCOL, CSV, true, 256>), grid, block, lds, st, a, ff);
        else hipLaunchKernelGGL((fdm_stream_kernel<NCOL, CSV, false, 256>), grid, block, lds, st, a, ff);
    } else {
        if (ff.full) hipLaunchKernelGGL((fdm_stream_kernel<NCOL, CSV, true, 1024>), grid, block, lds, st, a, ff);
        else hipLaunchKernelGGL((fdm_stream_kernel<NCOL, CSV, false, 1024>), grid, block, lds, st, a, ff);
    }
}

template <int NCOL>
void launch_ncol_asym(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff)
{
    const dim3 grid((unsigned)(a.nchunk * a.sys_count)), block((unsigned)ff.threads);
    const bool fused = (op == SMOQY_OP_MTM || op == SMOQY_OP_MMT);
    const size_t lds = sizeof(double2) * (size_t)a.N * (size_t)(a.Tc + 1) * (fused ? 2 : 1);
    switch (op) {
        case SMOQY_OP_M: hipLaunchKernelGGL((fdm_fast_asym_kernel<NCOL, SMOQY_OP_M>), grid, block, lds, st, a, ff); break;
        case SMOQY_OP_MT: hipLaunchKernelGGL((fdm_fast_asym_kernel<NCOL, SMOQY_OP_MT>), grid, block, lds, st, a, ff); break;
        case SMOQY_OP_MTM: hipLaunchKernelGGL((fdm_fast_asym_kernel<NCOL, SMOQY_OP_MTM>), grid, block, lds, st, a, ff); break;
        default: hipLaunchKernelGGL((fdm_fast_asym_kernel<NCOL, SMOQY_OP_MMT>), grid, block, lds, st, a, ff); break;
    }
}

template <int NCOL, bool CSV, bool CPLX = false>
void launch_ncol(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff)
{
    const dim3 grid((unsigned)(a.nchunk * a.sys_count)), block((unsigned)ff.threads);
    const size_t lds = sizeof(double2) * (size_t)a.N * (size_t)(a.Tc + 1);
    switch (op) {
        case SMOQY_OP_M: hipLaunchKernelGGL((fdm_fast_kernel<NCOL, SMOQY_OP_M, CSV, CPLX>), grid, block, lds, st, a, ff); break;
        case SMOQY_OP_MT: hipLaunchKernelGGL((fdm_fast_kernel<NCOL, SMOQY_OP_MT, CSV, CPLX>), grid, block, lds, st, a, ff); break;
        case SMOQY_OP_MTM: hipLaunchKernelGGL((fdm_fast_kernel<NCOL, SMOQY_OP_MTM, CSV, CPLX>), grid, block, lds, st, a, ff); break;
        default: hipLaunchKernelGGL((fdm_fast_kernel<NCOL, SMOQY_OP_MMT, CSV, CPLX>), grid, block, lds, st, a, ff); break;
    }
}

}  // namespace

bool fdm_fast_supported(const FdmArgs &a, const FdmFast &ff, bool sym)
{
    // the Asym kernel keeps a second LDS image for the fused products
    // complex hoppings (ff.csi): the Sym kernel only (round 4); Asym complex handles keep the generic kernel
    return ff.enabled && (sym || !ff.csi) && a.ncol >= 1 && a.ncol <= kFdmColours && a.Tc + 1 <= KMAX && sizeof(double2) * (size_t)a.N * (size_t)(a.Tc + 1) * (sym ? 1 : 2) <= 60 * 1024;
}

void launch_fdm_fast(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff, bool sym, bool cs_const)
{
    if (!sym) {
        switch (a.ncol) {
            case 1: launch_ncol_asym<1>(st, op, a, ff); break;
            case 2: launch_ncol_asym<2>(st, op, a, ff); break;
            case 3: launch_ncol_asym<3>(st, op, a, ff); break;
            default: launch_ncol_asym<4>(st, op, a, ff); break;
        }
        return;
    }
    if (ff.csi) {  // T = ComplexF64
        switch (a.ncol) {
            case 1: launch_ncol<1, true, true>(st, op, a, ff); break;
            case 2: launch_ncol<2, true, true>(st, op, a, ff); break;
            case 3: launch_ncol<3, true, true>(st, op, a, ff); break;
            default: launch_ncol<4, true, true>(st, op, a, ff); break;
        }
        return;
    }
    if (cs_const) {
        switch (a.ncol) {
            case 1: launch_ncol<1, false>(st, op, a, ff); break;
            case 2: launch_ncol<2, false>(st, op, a, ff); break;
            case 3: launch_ncol<3, false>(st, op, a, ff); break;
            default: launch_ncol<4, false>(st, op, a, ff); break;
        }
        return;
    }
    switch (a.ncol) {
        case 1: launch_ncol<1, true>(st, op, a, ff); break;
        case 2: launch_ncol<2, true>(st, op, a, ff); break;
        case 3: launch_ncol<3, true>(st, op, a, ff); break;
        default: launch_ncol<4, true>(st, op, a, ff); break;
    }
}

// streaming MᵀM: Sym, real hoppings, run length a multiple of the τ-chunk (the p·Ap partials keep the chunk layout), at least two slices
bool fdm_stream_supported(const FdmArgs &a, const FdmFast &ff, bool sym)
{
    return sym && ff.enabled && !ff.csi && a.ncol >= 1 && a.ncol <= kFdmColours && a.run_len >= 2 && a.run_len % a.Tc == 0 && a.Lt >= 4 && a.N <= 2 * ff.threads &&
           sizeof(double2) * 4 * (size_t)a.N <= 150 * 1024;
}

void launch_fdm_stream(hipStream_t st, const FdmArgs &a, const FdmFast &ff, bool cs_const)
{
    if (cs_const) {
        switch (a.ncol) {
            case 1: launch_stream_ncol<1, false>(st, a, ff); break;
            case 2: launch_stream_ncol<2, false>(st, a, ff); break;
            case 3: launch_stream_ncol<3, false>(st, a, ff); break;
            default: launch_stream_ncol<4, false>(st, a, ff); break;
        }
        return;
    }
    switch (a.ncol) {
        case 1: launch_stream_ncol<1, true>(st, a, ff); break;
        case 2: launch_stream_ncol<2, true>(st, a, ff); break;
        case 3: launch_stream_ncol<3, true>(st, a, ff); break;
        default: launch_stream_ncol<4, true>(st, a, ff); break;
    }
}

hipError_t configure_fdm_stream_kernels(const char **what)
{
    hipError_t first = hipSuccess;
    SMOQY_SET_LDS((fdm_stream_kernel<1, false, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, false, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, false, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, false, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, false, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, false, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, false, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, false, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, false, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, false, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, false, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, false, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, false, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, false, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, false, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, false, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, true, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, true, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, true, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<1, true, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, true, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, true, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, true, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<2, true, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, true, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, true, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, true, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<3, true, true, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, true, false, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, true, false, 1024>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, true, true, 256>), 160 * 1024 - 512);
    SMOQY_SET_LDS((fdm_stream_kernel<4, true, true, 1024>), 160 * 1024 - 512);
    return first;
}

// pack cosh/sinh into the padded interleaved table csf[l][idx] = (c, s) (self bonds: (1, 0)) and note
// per walker whether the hoppings depend on τ at all (Lt = nwalkers * Lt1 slices in a row)
__global__ void pack_csf_kernel(const double *__restrict__ ch, const double *__restrict__ sh, const int *__restrict__ psrc, double2 *__restrict__ csf, int *__restrict__ cs_varies, int Lt, int Lt1, int Nh,
                                int ptotal, const double *__restrict__ shi, double *__restrict__ csi)
{
    const size_t tot = (size_t)Lt * ptotal;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / ptotal), j = (int)(idx - (size_t)l * ptotal);
        const int h = psrc[j];
        double2 v = make_double2(1.0, 0.0);
        double im = 0.0;
        if (h >= 0) {
            v = make_double2(ch[(size_t)l * Nh + h], sh[(size_t)l * Nh + h]);
            const int w = l / Lt1, lf = w * Lt1;  // first slice of this walker
            bool differs = v.x != ch[(size_t)lf * Nh + h] || v.y != sh[(size_t)lf * Nh + h];
            if (shi) {
                im = shi[(size_t)l * Nh + h];
                differs = differs || im != shi[(size_t)lf * Nh + h];
            }
            if (differs) cs_varies[w] = 1;  // benign race: every writer stores 1
        }
        csf[idx] = v;
        if (csi) csi[idx] = im;
    }
}

// shi / csi: Im sinh per bond and its padded copy (T = ComplexF64), or nullptr
void launch_pack_csf(hipStream_t st, const double *ch, const double *sh, const int *psrc, double2 *csf, int *cs_varies, int Lt, int Lt1, int Nh, int ptotal, const double *shi, double *csi)
{
    const size_t tot = (size_t)Lt * ptotal;
    if (tot == 0) return;
    (void)hipMemsetAsync(cs_varies, 0, sizeof(int) * (size_t)(Lt / Lt1), st);
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_csf_kernel, dim3(blocks), dim3(256), 0, st, ch, sh, psrc, csf, cs_varies, Lt, Lt1, Nh, ptotal, shi, csi);
}

}  // namespace smoqy
