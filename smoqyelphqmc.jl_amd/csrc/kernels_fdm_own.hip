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
// CS1 = 0: one (cosh, sinh) pair per colour for every slot (τ-independent hoppings: only csx / csy[c][0] are set and read)
template <int NCOL, int KM, int SH, int CS1 = 1>
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
                const double c = ln.csx[Q][(k + SH) * CS1].x, s = ln.csx[Q][(k + SH) * CS1].y;                              \
                const double2 t = lino(c, ux[k], s, uy[k]);                                                 \
                uy[k] = lino(c, uy[k], s, ux[k]);                                                           \
                ux[k] = t;                                                                                  \
            }                                                                                               \
        } else {                                                                                            \
            double2 mx[KM], my[KM];                                                                         \
            FDM_OWN_EXCHANGE(c_, mx, my)                                                                    \
            _Pragma("unroll") for (int k = 0; k < KM - SH; ++k) if (k < nk) {                               \
                ux[k] = lino(ln.csx[c_][(k + SH) * CS1].x, ux[k], ln.csx[c_][(k + SH) * CS1].y, mx[k]);                     \
                uy[k] = lino(ln.csy[c_][(k + SH) * CS1].x, uy[k], ln.csy[c_][(k + SH) * CS1].y, my[k]);                     \
            }                                                                                               \
        }                                                                                                   \
    }
#pragma unroll
    for (int c = NCOL - 1; c >= 1; --c) FDM_OWN_STAGE(c)
    if (NCOL == 1) {  // C₁ D C₁ on the lane's own bond
#pragma unroll
        for (int k = 0; k < KM - SH; ++k)
            if (k < nk) {
                const double c = ln.csx[0][(k + SH) * CS1].x, s = ln.csx[0][(k + SH) * CS1].y;
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
                    const double c = ln.csx[0][(k + SH) * CS1].x, s = ln.csx[0][(k + SH) * CS1].y;
                    const double2 x = sclo(ln.dx[k + SH], lino(c, ux[k], s, mx[k])), xm = sclo(ln.dmx[k + SH], lino(c, mx[k], s, ux[k]));
                    ux[k] = lino(c, x, s, xm);
                }
                {
                    const double c = ln.csy[0][(k + SH) * CS1].x, s = ln.csy[0][(k + SH) * CS1].y;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// Streaming MᵀM on the owner-computes lane program (τ-independent hoppings: one (cosh, sinh) pair per colour).  Workgroup = (run of R output
// slices [la, lb), system), as in fdm_stream_kernel (kernels_fdm_fast.hip) — the same recurrences
//     P1(m):  y[m] = v[m] − h B_m v[m−1]          P2(m):  out[m−1] = y[m−1] − h̄ B_m y[m],      p·Ap = Σ |y|²,
// iteration j doing P1(j+1) and P2(j) as the two slots of ONE propagate — but the slices never enter LDS: lane t keeps v and y at the two
// sites of its colour-1 bond in registers (loaded straight from global memory, two slices ahead), a B apply is propagate_own — colour 1 in
// registers, colour 0 as DPP row rotations where FdmFast::wl0 allows, the other colours one LDS exchange each — i.e. TWO barriers per
// iteration on the honeycomb lattice where fdm_stream_kernel has five.  That kernel spends half its time in its stage chain at every launch
// size; this is the same pipeline with the shorter chain.
template <int NCOL, int TMAX>
__global__ void __launch_bounds__(TMAX) fdm_own_stream_kernel(FdmArgs a, FdmFast ff)
{
    extern __shared__ double2 U[];
    __shared__ double red[34];
    constexpr int KM = 2;
    const int Lt = a.Lt, N = a.N, T = blockDim.x, T2 = 2 * T, j = threadIdx.x;
    const int R = a.run_len, nrun = (Lt + R - 1) / R;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);  // XCD x: a contiguous share of the systems
    const int run = bid % nrun, sys = a.sys_first + bid / nrun;
    stamp_begin(a.stamp);
    const int sys_done = a.cg[sys].done;  // (a.cg is never null) acted on below, behind the first loads
    const int w = sys / a.nrhs;
    const int la = run * R, lb = min(Lt, la + R);
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double2 *csf = ff.csf + (size_t)w * Lt * ff.ptotal;
    double2 *W0 = U, *W1 = U + (size_t)KM * T2;

    // ---- the lane's program (clamped, unconditional loads: lanes beyond own_n compute on copies of lane 0's data and store nothing) ----
    OwnLane<NCOL, KM> ln;
    ln.on = j < ff.own_n;
    const int jc = ln.on ? j : 0;
    const int *own = ff.own;
    const int sx = own[jc], sy = own[T + jc], m0x = own[2 * T + jc], m0y = own[3 * T + jc];
    ln.ox = j; ln.oy = (sy != sx) ? T + j : j;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        ln.px[c] = own[(4 + 4 * c + 0) * T + jc];
        ln.py[c] = own[(4 + 4 * c + 1) * T + jc];
        const int bx = own[(4 + 4 * c + 2) * T + jc], by = own[(4 + 4 * c + 3) * T + jc];
        const double2 cx = csf[bx], cy = csf[by];  // slice 0 of the packed table: the hoppings do not depend on τ
        ln.csx[c][0] = cx;  // one pair per colour for both slots (propagate_own<…, CS1 = 0>)
        ln.csy[c][0] = cy;
    }
    if (!ln.on) { ln.px[0] = ln.py[0] = j; }
    // one slice at the lane's sites / exp(-ΔτV) of one slice at the lane's sites and their colour-0 mates (all unconditional)
#define OS_LOAD_V(vx_, vy_, m_)                                      \
    {                                                                \
        const double2 *row_ = in + (size_t)wrapo((m_), Lt) * sstride; \
        vx_ = row_[sx];                                              \
        vy_ = row_[sy];                                              \
    }
#define OS_LOAD_F(d0_, d1_, d2_, d3_, m_)                            \
    {                                                                \
        const double *e_ = expV + (size_t)wrapo((m_), Lt) * N;       \
        d0_ = e_[sx]; d1_ = e_[sy]; d2_ = e_[m0x]; d3_ = e_[m0y];    \
    }
    // what the loop waits for first goes out first: the two prefetched slices, then the fields of slice la+2, then the prologue's own data
    double2 pax, pay, pbx, pby;
    OS_LOAD_V(pax, pay, min(la + 2, lb))
    OS_LOAD_V(pbx, pby, min(la + 3, lb))
    asm volatile("" ::: "memory");  // keeps the loads above on this side of the early return
    if (sys_done) return;           // workgroup-uniform; nothing stored yet
    double fn0, fn1, fn2, fn3;      // fields of the slice after the two in use
    OS_LOAD_F(fn0, fn1, fn2, fn3, min(la + 2, lb))
    double f10, f11, f12, f13, f20, f21, f22, f23;  // slot 1: B_{m+1}, slot 2: B_m
    OS_LOAD_F(f20, f21, f22, f23, la)
    OS_LOAD_F(f10, f11, f12, f13, la + 1)
    double2 vmx, vmy, v0x, v0y, v1x, v1y;
    OS_LOAD_V(vmx, vmy, la - 1)
    OS_LOAD_V(v0x, v0y, la)
    OS_LOAD_V(v1x, v1y, la + 1)
    int buf = 0;
    double accr = 0.0;
    double2 ux[KM], uy[KM];
#define OS_SET_FIELDS()                                                          \
    {                                                                            \
        ln.dx[0] = f10; ln.dy[0] = f11; ln.dmx[0] = f12; ln.dmy[0] = f13;        \
        ln.dx[1] = f20; ln.dy[1] = f21; ln.dmx[1] = f22; ln.dmy[1] = f23;        \
    }
    // ---- prologue: B_{la+1} v[la] (slot 1) and B_la v[la−1] (slot 2) together -> y[la], y[la+1] ----
    OS_SET_FIELDS()
    ux[0] = v0x; uy[0] = v0y; ux[1] = vmx; uy[1] = vmy;
    propagate_own<NCOL, KM, 0, 0>(ln, ux, uy, 2, W0, W1, buf, T2, ff.wl0);
    double2 yprev_x = hopcomb_o(v0x, ux[1], wrapo(la, Lt) == 0, false, a), yprev_y = hopcomb_o(v0y, uy[1], wrapo(la, Lt) == 0, false, a);          // y[la]
    double2 ycur_x = hopcomb_o(v1x, ux[0], wrapo(la + 1, Lt) == 0, false, a), ycur_y = hopcomb_o(v1y, uy[0], wrapo(la + 1, Lt) == 0, false, a);    // y[la+1]
    double2 vcur_x = v1x, vcur_y = v1y;  // v[la+1]
    const bool two = sy != sx;
    if (ln.on) accr += yprev_x.x * yprev_x.x + yprev_x.y * yprev_x.y + (two ? yprev_y.x * yprev_y.x + yprev_y.y * yprev_y.y : 0.0);
    f20 = f10; f21 = f11; f22 = f12; f23 = f13;   // slot 2 of the first iteration: fields of slice la+1
    f10 = fn0; f11 = fn1; f12 = fn2; f13 = fn3;   // slot 1: fields of slice la+2
    // ---- steady state: iteration jj does P1(jj+1) (its result is unused at jj = lb) and P2(jj); written out twice per trip so that the two
    // prefetch register sets alternate without a copy of a register that is still in flight ----
#define OS_ITER(px_, py_, jj_)                                                                                        \
    {                                                                                                                 \
        const int jj = (jj_);                                                                                         \
        OS_LOAD_F(fn0, fn1, fn2, fn3, min(jj + 2, lb))                                                                \
        const double2 vnx = px_, vny = py_;  /* v[jj+1], requested two iterations ago */                              \
        OS_LOAD_V(px_, py_, min(jj + 3, lb))                                                                          \
        OS_SET_FIELDS()                                                                                               \
        ux[0] = vcur_x; uy[0] = vcur_y; ux[1] = ycur_x; uy[1] = ycur_y;                                               \
        propagate_own<NCOL, KM, 0, 0>(ln, ux, uy, 2, W0, W1, buf, T2, ff.wl0);                                           \
        if (ln.on) { /* out[jj−1] = y[jj−1] − h̄ B_jj y[jj] */                                                          \
            const bool wrap = (jj - 1) == Lt - 1;                                                                     \
            double2 *row = out + (size_t)(jj - 1) * sstride;                                                          \
            row[sx] = hopcomb_o(yprev_x, ux[1], wrap, true, a);                                                       \
            if (two) row[sy] = hopcomb_o(yprev_y, uy[1], wrap, true, a);                                              \
        }                                                                                                             \
        yprev_x = ycur_x; yprev_y = ycur_y;                                                                           \
        if (jj < lb) {                                                                                                \
            if (ln.on) accr += yprev_x.x * yprev_x.x + yprev_x.y * yprev_x.y + (two ? yprev_y.x * yprev_y.x + yprev_y.y * yprev_y.y : 0.0); \
            ycur_x = hopcomb_o(vnx, ux[0], wrapo(jj + 1, Lt) == 0, false, a);                                         \
            ycur_y = hopcomb_o(vny, uy[0], wrapo(jj + 1, Lt) == 0, false, a);                                         \
            vcur_x = vnx; vcur_y = vny;                                                                               \
            f20 = f10; f21 = f11; f22 = f12; f23 = f13;                                                               \
            f10 = fn0; f11 = fn1; f12 = fn2; f13 = fn3;                                                               \
        }                                                                                                             \
    }
    int jj0 = la + 1;
    for (; jj0 + 1 <= lb; jj0 += 2) {
        OS_ITER(pax, pay, jj0)
        OS_ITER(pbx, pby, jj0 + 1)
    }
    if (jj0 <= lb) OS_ITER(pax, pay, jj0)
#undef OS_ITER
#undef OS_SET_FIELDS
#undef OS_LOAD_F
#undef OS_LOAD_V
    if (a.partial) {
        for (int off = 32; off > 0; off >>= 1) accr += __shfl_down(accr, off, 64);
        const int wave = j >> 6, lane = j & 63, nwave = (T + 63) >> 6;
        __syncthreads();  // the LDS images are done with
        if (lane == 0) red[wave] = accr;
        __syncthreads();
        if (j == 0) {
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
    return tuning_env(kTuneFdmOwn) == 0 ? 0 : 1;  // A/B switch for measurements; default on
}

}  // namespace

// Measured on MI355X (honeycomb L = 16, Lτ = 128, fused MᵀM): 5.2 / 6.9 / 9.5 µs at 1 / 4 / 8 systems per launch against
// 6.6 / 8.2 / 11.1 µs for the LDS-resident kernel — the launch is a latency chain there and the shorter chain wins; from
// 16 systems on the chip is full either way and the LDS-resident kernel's higher occupancy wins (18.4 vs 18.9 µs at 16,
// 54.8 vs 60.3 µs at 64).  Hence the batch limit.
bool fdm_own_supported(const FdmArgs &a, const FdmFast &ff, bool sym)
{
    static const int own_max = tuning_env(kTuneFdmOwnMax) >= 0 ? tuning_env(kTuneFdmOwnMax) : 8;  // experiment knob: systems per launch up to which this kernel is chosen
    return sym && ff.enabled && !ff.csi && ff.own && fdm_own_enabled() && a.sys_count <= own_max && a.ncol >= 1 && a.ncol <= kFdmColours && a.Tc + 1 <= 3 &&
           sizeof(double2) * 2 * (size_t)(a.Tc + 1 <= 2 ? 2 : 3) * 2 * (size_t)ff.threads <= 64 * 1024;
}

// streaming MᵀM on the lane program: τ-independent hoppings (the caller passes cs_const), Sym, 256-lane lattices, R >= 2
bool fdm_own_stream_supported(const FdmArgs &a, const FdmFast &ff, bool sym, bool cs_const)
{
    // for handles that carry 32 systems or more (SMOQY_FDM_OWNSTREAM=1: always, 0: never).  Alone on the device it is the shorter kernel at every size —
    // 13.6 / 22.1 / 37.1 / 76.3 us against fdm_stream_kernel's 15.0 / 24.3 / 42.0 / 81.2 at 16 / 32 / 64 / 128 systems — but its 152 VGPRs
    // (three waves per SIMD) cost what the shorter stage chain gains once other solves' kernels share the device: with eight batches of 16
    // in flight (the bench) nothing is gained and its launches are longer, while one stream of 32 / 64 walkers gains 1-3 % per sweep, a
    // 64-member team 3.7 %, and four teams of 32 or eight batches of 32 are unchanged.  Hence the rule by handle size.
    static const int mode = tuning_env(kTuneFdmOwnStream) < 0 ? -1 : (tuning_env(kTuneFdmOwnStream) == 1 ? 1 : 0);
    const bool on = mode == 1 || (mode < 0 && a.nsys >= 32);  // the handle's systems, whatever part of them this launch covers (smoqy_cg_split)
    return on && sym && cs_const && ff.enabled && ff.own && ff.threads <= 256 && a.ncol >= 2 && a.ncol <= kFdmColours && a.run_len >= 2 && a.run_len % a.Tc == 0 &&
           a.Lt >= 4 && a.shi == nullptr;
}

void launch_fdm_own_stream(hipStream_t st, const FdmArgs &a, const FdmFast &ff)
{
    const int nrun = (a.Lt + a.run_len - 1) / a.run_len;
    const dim3 grid((unsigned)(nrun * a.sys_count)), block((unsigned)ff.threads);
    const size_t lds = sizeof(double2) * 2 * 2 * 2 * (size_t)ff.threads;  // two images of KM = 2 slots x 2T values
    switch (a.ncol) {
        case 2: hipLaunchKernelGGL((fdm_own_stream_kernel<2, 256>), grid, block, lds, st, a, ff); break;
        case 3: hipLaunchKernelGGL((fdm_own_stream_kernel<3, 256>), grid, block, lds, st, a, ff); break;
        default: hipLaunchKernelGGL((fdm_own_stream_kernel<4, 256>), grid, block, lds, st, a, ff); break;
    }
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
