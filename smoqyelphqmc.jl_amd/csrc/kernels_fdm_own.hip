// Owner-computes FermionDetMatrix kernels for gfx950 (Sym, <= kFdmColours colours, tau-chunk <= 2):
// M, Mᵀ, MᵀM, MMᵀ as ONE launch each.  Reference semantics: src/FermionDetMatrix.jl:385-427, 484-525,
// 329-340, 357-368 and src/checkerboard_matrix_multiply.jl:50-69, exactly as kernels_fdm_fast.hip.
//
// kernels_fdm_fast.hip keeps the propagating slices in LDS and pays one LDS read-modify-write plus a
// workgroup barrier per colour stage (2L-1 per B apply), a staging pass of the input through LDS, and a
// hand-over pass between the two halves of the fused MᵀM.  Here lane j OWNS the two sites of the j-th
// padded bond of colour q = 1 for the whole kernel and carries them — for all slices of the chunk — in
// registers:
//   * its values come straight from global memory and go straight back (no staging pass);
//   * a stage of the owned colour is register arithmetic; a stage of another colour is one exchange
//     (own values to an LDS image, barrier, read the two mates) and  a' = c·a + s·mate(a)  per own site;
//     q occurs twice in  C_L…C_2 (C_1 D C_1) C_2…C_L,  so B costs 2L-3 exchanges instead of 2L-1 stages
//     (3 instead of 5 on the honeycomb lattice, 1 instead of 3 on a chain);
//   * the fused centre stage recomputes the mate's intermediate value (same bond, mate's exp(-ΔτV));
//   * "v ∓ B v", the hand-over from M to Mᵀ and the dot(in, out) partial act on the lane's own registers.
// Exchanges ping-pong between two LDS images, so each costs one barrier.  Per-site arithmetic (order of
// operations included) is that of kernels_fdm_fast.hip.
#include "smoqy_internal.h"

#include <cstdlib>

namespace smoqy {

namespace {

__device__ __forceinline__ int wrapo(int l, int Lt) { return l >= Lt ? l - Lt : (l < 0 ? l + Lt : l); }
__device__ __forceinline__ double2 lino(double a, double2 x, double b, double2 y) { return make_double2(a * x.x + b * y.x, a * x.y + b * y.y); }
__device__ __forceinline__ double2 sclo(double a, double2 x) { return make_double2(a * x.x, a * x.y); }
__device__ __forceinline__ double2 hopcomb_o(double2 v, double2 u, bool wrap, bool dagger, const FdmArgs &a)
{
    double pr = a.hop_re, pi = dagger ? -a.hop_im : a.hop_im;
    if (wrap && a.antiperiodic) { pr = -pr; pi = -pi; }
    return make_double2(v.x - (pr * u.x - pi * u.y), v.y - (pr * u.y + pi * u.x));
}

// rotation within rows of 16 lanes as a DPP modifier (see kernels_kpm.hip, row_rot): CTRL = 0x120 + n is row_ror:n
template <int CTRL>
__device__ __forceinline__ double2 row_rot_o(double2 x)
{
    int a0 = __double2loint(x.x), a1 = __double2hiint(x.x), b0 = __double2loint(x.y), b1 = __double2hiint(x.y);
    a0 = __builtin_amdgcn_update_dpp(0, a0, CTRL, 0xf, 0xf, false);
    a1 = __builtin_amdgcn_update_dpp(0, a1, CTRL, 0xf, 0xf, false);
    b0 = __builtin_amdgcn_update_dpp(0, b0, CTRL, 0xf, 0xf, false);
    b1 = __builtin_amdgcn_update_dpp(0, b1, CTRL, 0xf, 0xf, false);
    return make_double2(__hiloint2double(a1, a0), __hiloint2double(b1, b0));
}

template <int NCOL, int KM>
struct OwnLane {
    int ox, oy;                   // LDS slots of the two own sites (equal for a self bond)
    int px[NCOL], py[NCOL];       // slots of their mates in colour c
    double2 csx[NCOL][KM], csy[NCOL][KM];  // (cosh, sinh) of the colour-c bond at own site x / y on field slice k
    double dx[KM], dy[KM], dmx[KM], dmy[KM];  // exp(-ΔτV) at the own sites and at their colour-0 mates
    bool on;
};

// (ux, uy)[k] <- B (ux, uy)[k] for k < nk, fields of register slice k + SH
template <int NCOL, int KM, int SH>
__device__ __forceinline__ void propagate_own(const OwnLane<NCOL, KM> &ln, double2 (&ux)[KM], double2 (&uy)[KM], int nk, double2 *W0, double2 *W1, int &buf, int T2, int wl0)
{
    constexpr int Q = NCOL >= 2 ? 1 : 0;
#define FDM_OWN_EXCHANGE(c_, mx_, my_)                                              \
    {                                                                               \
        double2 *Wc = buf ? W1 : W0;                                                \
        buf ^= 1;                                                                   \
        if (ln.on) {                                                                \
            _Pragma("unroll") for (int k = 0; k < KM - SH; ++k) if (k < nk) {       \
                Wc[k * T2 + ln.ox] = ux[k];                                         \
                Wc[k * T2 + ln.oy] = uy[k];                                         \
            }                                                                       \
        }                                                                           \
        __syncthreads();                                                            \
        _Pragma("unroll") for (int k = 0; k < KM - SH; ++k) if (k < nk) {           \
            mx_[k] = Wc[k * T2 + ln.px[c_]];                                        \
            my_[k] = Wc[k * T2 + ln.py[c_]];                                        \
        }                                                                           \
    }
#define FDM_OWN_STAGE(c_)                                                                                   \
    {                                                                                                       \
        if ((c_) == Q) {                                                                                    \
            _Pragma("unroll") for (int k = 0; k < KM - SH; ++k) if (k < nk) {                               \
                const double c = ln.csx[Q][k + SH].x, s = ln.csx[Q][k + SH].y;                              \
                const double2 t = lino(c, ux[k], s, uy[k]);                                                 \
                uy[k] = lino(c, uy[k], s, ux[k]);                                                           \
                ux[k] = t;                                                                                  \
            }                                                                                               \
        } else {                                                                                            \
            double2 mx[KM], my[KM];                                                                         \
            FDM_OWN_EXCHANGE(c_, mx, my)                                                                    \
            _Pragma("unroll") for (int k = 0; k < KM - SH; ++k) if (k < nk) {                               \
                ux[k] = lino(ln.csx[c_][k + SH].x, ux[k], ln.csx[c_][k + SH].y, mx[k]);                     \
                uy[k] = lino(ln.csy[c_][k + SH].x, uy[k], ln.csy[c_][k + SH].y, my[k]);                     \
            }                                                                                               \
        }                                                                                                   \
    }
#pragma unroll
    for (int c = NCOL - 1; c >= 1; --c) FDM_OWN_STAGE(c)
    if (NCOL == 1) {  // C₁ D C₁ on the lane's own bond
#pragma unroll
        for (int k = 0; k < KM - SH; ++k)
            if (k < nk) {
                const double c = ln.csx[0][k + SH].x, s = ln.csx[0][k + SH].y;
                const double2 x = sclo(ln.dx[k + SH], lino(c, ux[k], s, uy[k])), y = sclo(ln.dy[k + SH], lino(c, uy[k], s, ux[k]));
                ux[k] = lino(c, x, s, y);
                uy[k] = lino(c, y, s, x);
            }
    } else {          // one exchange; the mate's value after C₁ and D is recomputed (same bond, its own exp(-ΔτV))
        double2 mx[KM], my[KM];
        if (wl0 == 2) {         // the mates are the lane's row neighbours (FdmFast::wl0): two DPP row rotations, no LDS image, no barrier
#pragma unroll
            for (int k = 0; k < KM - SH; ++k) { mx[k] = row_rot_o<0x121>(uy[k]); my[k] = row_rot_o<0x12F>(ux[k]); }
        } else if (wl0 == 3) {
#pragma unroll
            for (int k = 0; k < KM - SH; ++k) { mx[k] = row_rot_o<0x12F>(uy[k]); my[k] = row_rot_o<0x121>(ux[k]); }
        } else {
            FDM_OWN_EXCHANGE(0, mx, my)
        }
#pragma unroll
        for (int k = 0; k < KM - SH; ++k)
            if (k < nk) {
                {
                    const double c = ln.csx[0][k + SH].x, s = ln.csx[0][k + SH].y;
                    const double2 x = sclo(ln.dx[k + SH], lino(c, ux[k], s, mx[k])), xm = sclo(ln.dmx[k + SH], lino(c, mx[k], s, ux[k]));
                    ux[k] = lino(c, x, s, xm);
                }
                {
                    const double c = ln.csy[0][k + SH].x, s = ln.csy[0][k + SH].y;
                    const double2 y = sclo(ln.dy[k + SH], lino(c, uy[k], s, my[k])), ym = sclo(ln.dmy[k + SH], lino(c, my[k], s, uy[k]));
                    uy[k] = lino(c, y, s, ym);
                }
            }
    }
#pragma unroll
    for (int c = 1; c <= NCOL - 1; ++c) FDM_OWN_STAGE(c)
#undef FDM_OWN_STAGE
#undef FDM_OWN_EXCHANGE
}

// TMAX bounds the workgroup size at compile time: lattices of up to 512 sites run with 256 lanes and the register
// budget of one wavefront per SIMD-quarter, larger ones with the 128-register budget of a 1024-lane workgroup
template <int NCOL, int OP, int KM, int TMAX>
__global__ void __launch_bounds__(TMAX) fdm_own_kernel(FdmArgs a, FdmFast ff)
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
    const int sys_done = a.cg[sys].done;  // (a.cg is never null)  // acted on below, with the lane program's loads already in flight (one round trip less in front of the data)
    const int w = sys / a.nrhs;
    const int Lt = a.Lt, N = a.N, T = blockDim.x, T2 = 2 * T, j = threadIdx.x;
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
    const int ubase = (OP == SMOQY_OP_M || OP == SMOQY_OP_MTM) ? l0 - 1 : (OP == SMOQY_OP_MT ? l0 + 1 : l0);  // source slice of u[0]
    const int vbase = (OP == SMOQY_OP_MMT) ? l0 - 1 : l0;      // slice of v[0]
    constexpr int VSH = (OP == SMOQY_OP_M || OP == SMOQY_OP_MTM) ? 1 : -1;  // v[k] is u[k + VSH]
    double2 *W0 = U, *W1 = U + (size_t)KM * T2;

    // ---- the lane's program and everything it needs from memory, requested up front ----
    OwnLane<NCOL, KM> ln;
    ln.on = j < ff.own_n;
    ln.ox = ln.oy = j;
    int sx = 0, sy = 0, m0x = 0, m0y = 0;
    int bix[NCOL], biy[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) { ln.px[c] = ln.py[c] = j; bix[c] = biy[c] = 0; }
    const int *own = ff.own;
    if (ln.on) {
        sx = own[j]; sy = own[T + j];
        m0x = own[2 * T + j]; m0y = own[3 * T + j];
        ln.oy = (sy != sx) ? T + j : j;
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            ln.px[c] = own[(4 + 4 * c + 0) * T + j];
            ln.py[c] = own[(4 + 4 * c + 1) * T + j];
            bix[c] = own[(4 + 4 * c + 2) * T + j];
            biy[c] = own[(4 + 4 * c + 3) * T + j];
        }
    }
    // hoppings that do not depend on τ are fetched once instead of once per slice
    const bool cs_varies = ff.cs_varies[w] != 0;
    asm volatile("" ::: "memory");  // compiler fence: keeps the loads above on this side of the early return (no instruction, no wait)
    if (sys_done) return;  // workgroup-uniform
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            ln.csx[c][k] = ln.csy[c][k] = make_double2(1.0, 0.0);
            if (ln.on && k < K1 && (k == 0 || cs_varies)) {
                const double2 *row = csf + (cs_varies ? (size_t)wrapo(fbase + k, Lt) : (size_t)0) * ff.ptotal;  // τ-independent hoppings: slice 0 for everyone (cache-resident)
                ln.csx[c][k] = row[bix[c]];
                ln.csy[c][k] = row[biy[c]];
            }
        }
        if (!cs_varies) {
#pragma unroll
            for (int k = 1; k < KM; ++k) { ln.csx[c][k] = ln.csx[c][0]; ln.csy[c][k] = ln.csy[c][0]; }
        }
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        ln.dx[k] = ln.dy[k] = ln.dmx[k] = ln.dmy[k] = 1.0;
        if (k < K1 && ln.on) {
            const double *e = expV + (size_t)wrapo(fbase + k, Lt) * N;
            ln.dx[k] = e[sx];
            ln.dy[k] = e[sy];
            if (NCOL >= 2) { ln.dmx[k] = e[m0x]; ln.dmy[k] = e[m0y]; }
        }
    }
    double2 ux[KM], uy[KM], vx[KM], vy[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        ux[k] = uy[k] = make_double2(0.0, 0.0);
        if (k < K1 && ln.on) {
            const double2 *row = in + (size_t)wrapo(ubase + k, Lt) * sstride;
            ux[k] = row[sx];
            uy[k] = row[sy];
        }
    }
    // the "v" of v ∓ B v: all but one of its slices are among the u slices (shifted by one)
#pragma unroll
    for (int k = 0; k < KM; ++k) {
        vx[k] = vy[k] = make_double2(0.0, 0.0);
        const int ku = k + VSH;
        if (k < K1) {
            if (ku >= 0 && ku < K1 && ku < KM) {
                vx[k] = ux[ku < 0 ? 0 : (ku >= KM ? KM - 1 : ku)];
                vy[k] = uy[ku < 0 ? 0 : (ku >= KM ? KM - 1 : ku)];
            } else if (ln.on) {
                const double2 *row = in + (size_t)wrapo(vbase + k, Lt) * sstride;
                vx[k] = row[sx];
                vy[k] = row[sy];
            }
        }
    }

    int buf = 0;
    double2 acc = make_double2(0.0, 0.0);
    propagate_own<NCOL, KM, 0>(ln, ux, uy, K1, W0, W1, buf, T2, ff.wl0);
    if (!FUSED) {
        // M:  out[l] = v[l] ∓ B_l v[l-1]  (+ on the first slice);  Mᵀ: out[l] = v[l] ∓ B_{l+1} v[l+1]  (+ on the last)
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k < nk && ln.on) {
                const int l = l0 + k;
                const bool wrap = (OP == SMOQY_OP_M) ? (l == 0) : (l == Lt - 1);
                const double2 oi = hopcomb_o(vx[k], ux[k], wrap, OP == SMOQY_OP_MT, a), oj = hopcomb_o(vy[k], uy[k], wrap, OP == SMOQY_OP_MT, a);
                double2 *row = out + (size_t)l * sstride;
                row[sx] = oi;
                acc.x += vx[k].x * oi.x + vx[k].y * oi.y;
                acc.y += vx[k].x * oi.y - vx[k].y * oi.x;
                if (sy != sx) {
                    row[sy] = oj;
                    acc.x += vy[k].x * oj.x + vy[k].y * oj.y;
                    acc.y += vy[k].x * oj.y - vy[k].y * oj.x;
                }
            }
        }
    } else {
        // y = first operator applied on nk+1 slices, in registers at the lane's own sites
        double2 yx[KM], yy[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            yx[k] = yy[k] = make_double2(0.0, 0.0);
            if (k < K1) {
                const int l = wrapo(vbase + k, Lt);  // slice of y[k]
                const bool wrap = (OP == SMOQY_OP_MTM) ? (l == 0) : (l == Lt - 1);
                yx[k] = hopcomb_o(vx[k], ux[k], wrap, OP == SMOQY_OP_MMT, a);
                yy[k] = hopcomb_o(vy[k], uy[k], wrap, OP == SMOQY_OP_MMT, a);
            }
        }
        // second operator: MᵀM propagates y[1..nk] with the fields of slices l0+1.., MMᵀ propagates y[0..nk-1]
        // with the fields of slices l0..  — the hand-over is a register rename
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const int ks = (OP == SMOQY_OP_MTM) ? k + 1 : k;
            ux[k] = (ks < KM) ? yx[ks < KM ? ks : 0] : make_double2(0.0, 0.0);
            uy[k] = (ks < KM) ? yy[ks < KM ? ks : 0] : make_double2(0.0, 0.0);
        }
        if (OP == SMOQY_OP_MTM) propagate_own<NCOL, KM, 1>(ln, ux, uy, nk, W0, W1, buf, T2, ff.wl0);
        else propagate_own<NCOL, KM, 0>(ln, ux, uy, nk, W0, W1, buf, T2, ff.wl0);
#pragma unroll
        for (int k = 0; k < KM - 1; ++k) {
            if (k < nk && ln.on) {
                const int l = l0 + k;
                const bool wrap = (OP == SMOQY_OP_MTM) ? (l == Lt - 1) : (l == 0);
                const double2 bi = (OP == SMOQY_OP_MTM) ? yx[k] : yx[k + 1], bj = (OP == SMOQY_OP_MTM) ? yy[k] : yy[k + 1];
                const double2 oi = hopcomb_o(bi, ux[k], wrap, OP == SMOQY_OP_MTM, a), oj = hopcomb_o(bj, uy[k], wrap, OP == SMOQY_OP_MTM, a);
                const double2 pi = (OP == SMOQY_OP_MTM) ? vx[k] : vx[k + 1], pj = (OP == SMOQY_OP_MTM) ? vy[k] : vy[k + 1];
                double2 *row = out + (size_t)l * sstride;
                row[sx] = oi;
                acc.x += pi.x * oi.x + pi.y * oi.y;
                acc.y += pi.x * oi.y - pi.y * oi.x;
                if (sy != sx) {
                    row[sy] = oj;
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

template <int NCOL, int KM>
void launch_own(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff)
{
    const dim3 grid((unsigned)(a.nchunk * a.sys_count)), block((unsigned)ff.threads);
    const size_t lds = sizeof(double2) * 2 * (size_t)KM * 2 * (size_t)ff.threads;
#define OWN_GO(OP_, TM_) hipLaunchKernelGGL((fdm_own_kernel<NCOL, OP_, KM, TM_>), grid, block, lds, st, a, ff)
    if (ff.threads <= 256) {
        switch (op) {
            case SMOQY_OP_M: OWN_GO(SMOQY_OP_M, 256); break;
            case SMOQY_OP_MT: OWN_GO(SMOQY_OP_MT, 256); break;
            case SMOQY_OP_MTM: OWN_GO(SMOQY_OP_MTM, 256); break;
            default: OWN_GO(SMOQY_OP_MMT, 256); break;
        }
    } else {
        switch (op) {
            case SMOQY_OP_M: OWN_GO(SMOQY_OP_M, 1024); break;
            case SMOQY_OP_MT: OWN_GO(SMOQY_OP_MT, 1024); break;
            case SMOQY_OP_MTM: OWN_GO(SMOQY_OP_MTM, 1024); break;
            default: OWN_GO(SMOQY_OP_MMT, 1024); break;
        }
    }
#undef OWN_GO
}

template <int NCOL>
void launch_own_km(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff)
{
    if (a.Tc + 1 <= 2) launch_own<NCOL, 2>(st, op, a, ff);
    else launch_own<NCOL, 3>(st, op, a, ff);
}

int fdm_own_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("SMOQY_FDM_OWN");  // A/B switch for measurements; default on
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

}  // namespace

// Measured on MI355X (honeycomb L = 16, Lτ = 128, fused MᵀM): 5.2 / 6.9 / 9.5 µs at 1 / 4 / 8 systems per launch against
// 6.6 / 8.2 / 11.1 µs for the LDS-resident kernel — the launch is a latency chain there and the shorter chain wins; from
// 16 systems on the chip is full either way and the LDS-resident kernel's higher occupancy wins (18.4 vs 18.9 µs at 16,
// 54.8 vs 60.3 µs at 64).  Hence the batch limit.
bool fdm_own_supported(const FdmArgs &a, const FdmFast &ff, bool sym)
{
    static const int own_max = [] { const char *e = getenv("SMOQY_FDM_OWN_MAX"); return e ? atoi(e) : 8; }();  // experiment knob: systems per launch up to which this kernel is chosen
    return sym && ff.enabled && ff.own && fdm_own_enabled() && a.sys_count <= own_max && a.ncol >= 1 && a.ncol <= kFdmColours && a.Tc + 1 <= 3 &&
           sizeof(double2) * 2 * (size_t)(a.Tc + 1 <= 2 ? 2 : 3) * 2 * (size_t)ff.threads <= 64 * 1024;
}

void launch_fdm_own(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff)
{
    switch (a.ncol) {
        case 1: launch_own_km<1>(st, op, a, ff); break;
        case 2: launch_own_km<2>(st, op, a, ff); break;
        case 3: launch_own_km<3>(st, op, a, ff); break;
        default: launch_own_km<4>(st, op, a, ff); break;
    }
}

}  // namespace smoqy
