// Imaginary-time FFT for the slice-major device layout, fused with the CG vector updates (gfx950).
//
// Reference: FourierTransformer (src/FourierTransformer.jl:39-64) inside the KPM preconditioner
// (src/KPMPreconditioner.jl:375, 406) and the BLAS-1 lines of cg_solve!
// (src/IterativeSolvers/ConjugateGradient.jl:219-245).
//
// Layout v[l][s][i]: a workgroup owns a tile of SB consecutive sites of one system for ALL Lτ
// slices (SB·16 = 128/256 contiguous bytes per slice), keeps it in LDS and runs a Stockham
// auto-sort FFT over τ with radix 2/3/4/5/7/8 passes (ping-pong between two LDS images; lanes run
// over (butterfly, site) with the site index fastest, so every LDS access is a run of whole
// 256-byte bank rows — conflict free).  Because a tile holds every τ of its sites, everything that
// is elementwise around the transform is fused into the same pass over memory:
//
//   forward  (MODE_FWD_CG):  α = (r·z)/(p·Ap);  r̂ -= α FFT(Ap);  partial |r|² (Parseval)
//   inverse  (MODE_INV_CG):  stop test on |r|/|b|;  x += α p;  β = (r·z)new/(r·z)old;  p = FFT⁻¹ ẑ + β p
// The fused CG keeps the residual in FREQUENCY space only: the transform is linear, so r̂ ← r̂ - α·FFT(Ap) replaces
// "update r, then transform it", the time-domain residual is never stored, and the x update moves next to the p update, which
// reads the old search direction anyway.  One iteration then touches 12 vectors (MᵀM 2, forward 3, Chebyshev 2 out of place,
// inverse 5) instead of 14; |r|² comes from Σ|r̂|²/Lτ and r·z from the Chebyshev kernel, both by Parseval.
//
// The CG runs in the twiddled basis (kernels_vec.hip), so no θ factors appear here; the plain
// modes take optional pre/post twiddle tables for the stand-alone FourierTransformer API.
// Lengths with a prime factor above 7 stay on rocFFT.
#include "smoqy_internal.h"

#include <cstdlib>

namespace smoqy {

namespace {

__device__ __forceinline__ double2 cm(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) or +i (inverse)
__device__ __forceinline__ double2 mul_mi(double2 a, bool inv) { return inv ? make_double2(-a.y, a.x) : make_double2(a.y, -a.x); }

// every kernel of this file is launched with 256 lanes.  A compile-time constant, not blockDim.x: that is a 16-bit global load from the
// dispatch packet, and inside the pass loops the compiler re-issued it in every pass with an s_waitcnt vmcnt(0) behind it — a memory round
// trip per pass on the critical path, and a drain of whatever loads were meant to travel under the passes
constexpr int kTfftThreads = 256;

template <int R>
__device__ __forceinline__ void dft(double2 (&v)[R], const double2 *__restrict__ wt, int Lt, bool inv)
{
    double2 o[R];
    const int step = Lt / R;
#pragma unroll
    for (int s = 0; s < R; ++s) {
        double2 acc = v[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            double2 w = wt[((r * s) % R) * step];
            if (inv) w.y = -w.y;
            acc = cadd(acc, cm(v[r], w));
        }
        o[s] = acc;
    }
#pragma unroll
    for (int s = 0; s < R; ++s) v[s] = o[s];
}

template <>
__device__ __forceinline__ void dft<2>(double2 (&v)[2], const double2 *__restrict__, int, bool)
{
    const double2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft<4>(double2 (&v)[4], const double2 *__restrict__, int, bool inv)
{
    const double2 a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]), inv);
    v[0] = cadd(a, c);
    v[1] = cadd(b, d);
    v[2] = csub(a, c);
    v[3] = csub(b, d);
}

// radix 5 written out (two real constants per conjugate pair instead of the table-driven sum: what lets the in-place form take
// lengths with a factor 5 without the register count of the generic butterfly)
template <>
__device__ __forceinline__ void dft<5>(double2 (&v)[5], const double2 *__restrict__, int, bool inv)
{
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos 72°, cos 144°
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin 72°, sin 144°
    const double2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const double2 m1 = make_double2(v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y);
    const double2 m2 = make_double2(v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y);
    // forward: X_k = m − i (…); inverse: + i (…)
    const double2 u1 = mul_mi(make_double2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y), inv);
    const double2 u2 = mul_mi(make_double2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y), inv);
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(m1, u1);
    v[4] = csub(m1, u1);
    v[2] = cadd(m2, u2);
    v[3] = csub(m2, u2);
}

template <>
__device__ __forceinline__ void dft<8>(double2 (&v)[8], const double2 *__restrict__, int, bool inv)
{
    // two radix-4 transforms on the even / odd inputs, then one radix-2 stage with the 8th roots
    double2 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft<4>(e, nullptr, 0, inv);
    dft<4>(o, nullptr, 0, inv);
    const double h = 0.70710678118654752440;
    // w8^1 = (1 -/+ i)/√2, w8^2 = -/+ i, w8^3 = (-1 -/+ i)/√2   (forward / inverse)
    const double2 t1 = inv ? make_double2(h * (o[1].x - o[1].y), h * (o[1].x + o[1].y)) : make_double2(h * (o[1].x + o[1].y), h * (o[1].y - o[1].x));
    const double2 t2 = mul_mi(o[2], inv);
    const double2 t3 = inv ? make_double2(-h * (o[3].x + o[3].y), h * (o[3].x - o[3].y)) : make_double2(h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y));
    v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
    v[1] = cadd(e[1], t1);   v[5] = csub(e[1], t1);
    v[2] = cadd(e[2], t2);   v[6] = csub(e[2], t2);
    v[3] = cadd(e[3], t3);   v[7] = csub(e[3], t3);
}

// one Stockham pass of radix R: in/out are LDS images [Lt][SB]
template <int R>
__device__ __forceinline__ void stockham_pass(const double2 *__restrict__ in, double2 *__restrict__ out, const double2 *__restrict__ wt, int Lt, int SB, int Ns, bool inv)
{
    const int stride = Lt / R, nb = stride * SB, tw_step = Lt / (Ns * R);
    for (int idx = threadIdx.x; idx < nb; idx += kTfftThreads) {
        const int j = idx / SB, sb = idx - j * SB;
        const int k = j % Ns;
        double2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double2 x = in[(size_t)(j + r * stride) * SB + sb];
            if (r > 0 && k > 0) {
                double2 w = wt[r * k * tw_step];
                if (inv) w.y = -w.y;
                x = cm(x, w);
            }
            v[r] = x;
        }
        dft<R>(v, wt, Lt, inv);
        const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[(size_t)(j0 + r * Ns) * SB + sb] = v[r];
    }
    __syncthreads();
}

// runs all passes; returns the buffer that holds the result
__device__ __forceinline__ double2 *stockham(double2 *A, double2 *B, const double2 *wt, const TfftArgs &a, bool inv)
{
    int Ns = 1;
    double2 *src = A, *dst = B;
    for (int f = 0; f < a.nfac; ++f) {
        const int R = (int)((a.fpack >> (4 * f)) & 15u);  // not a.fac[f]: a dynamically indexed kernel argument is a global load + wait in every pass
        switch (R) {
            case 2: stockham_pass<2>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
            case 3: stockham_pass<3>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
            case 4: stockham_pass<4>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
            case 5: stockham_pass<5>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
            case 7: stockham_pass<7>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
            default: stockham_pass<8>(src, dst, wt, a.Lt, a.SB, Ns, inv); break;
        }
        Ns *= R;
        double2 *t = src; src = dst; dst = t;
    }
    return src;
}

// ---- in-place form: one LDS image.  A butterfly reads and writes the same R rows, so a pass needs no second buffer and no
// read-all-before-write barrier; the digit-reversed order this leaves behind is absorbed where the tile meets global memory
// (forward: spectrum element k is read from row pos[k]; inverse: it is staged into row pos[k]).
// DIF (forward): y_s[j] = w_n^{js} Σ_q x[j + q m] w_R^{qs} stored at row s·m + j; DIT (inverse) undoes exactly that.
template <int R, bool DIT>
__device__ __forceinline__ void inplace_pass(double2 *__restrict__ X, const double2 *__restrict__ wt, int Lt, int SB, int ncur, bool inv)
{
    const int m = ncur / R, nb = (Lt / R) * SB, tws = Lt / ncur, stride = m * SB;
    for (int idx = threadIdx.x; idx < nb; idx += kTfftThreads) {
        const int jg = idx / SB, sb = idx - jg * SB;
        const int blk = jg / m, j = jg - blk * m;
        double2 *x = X + (size_t)(blk * ncur + j) * SB + sb;
        double2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = x[q * stride];
        if (DIT && j > 0) {
#pragma unroll
            for (int s = 1; s < R; ++s) {
                double2 w = wt[j * s * tws];
                if (inv) w.y = -w.y;
                v[s] = cm(v[s], w);
            }
        }
        dft<R>(v, wt, Lt, inv);
        if (!DIT && j > 0) {
#pragma unroll
            for (int s = 1; s < R; ++s) {
                double2 w = wt[j * s * tws];
                if (inv) w.y = -w.y;
                v[s] = cm(v[s], w);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) x[q * stride] = v[q];
    }
    __syncthreads();
}

template <bool DIT>
__device__ __forceinline__ void inplace_one(double2 *X, const double2 *wt, const TfftArgs &a, int R, int ncur, bool inv)
{
    switch (R) {
        case 2: inplace_pass<2, DIT>(X, wt, a.Lt, a.SB, ncur, inv); break;
        case 3: inplace_pass<3, DIT>(X, wt, a.Lt, a.SB, ncur, inv); break;
        case 5: inplace_pass<5, DIT>(X, wt, a.Lt, a.SB, ncur, inv); break;  // written-out butterfly (the table-driven radix 7 would set the register count: such lengths keep the two-image form)
        default: inplace_pass<4, DIT>(X, wt, a.Lt, a.SB, ncur, inv); break;
    }
}

// forward: natural order in, spectrum element k at row pos[k] out
__device__ __forceinline__ void inplace_forward(double2 *X, const double2 *wt, const TfftArgs &a)
{
    int ncur = a.Lt;
    for (int f = 0; f < a.snfac; ++f) {
        const int R = (int)((a.sfpack >> (4 * f)) & 15u);
        inplace_one<false>(X, wt, a, R, ncur, false);
        ncur /= R;
    }
}

// inverse: spectrum element k at row pos[k] in, natural order out (unnormalised)
__device__ __forceinline__ void inplace_inverse(double2 *X, const double2 *wt, const TfftArgs &a)
{
    int ncur = 1;
    for (int f = a.snfac - 1; f >= 0; --f) {
        const int R = (int)((a.sfpack >> (4 * f)) & 15u);
        ncur *= R;
        inplace_one<true>(X, wt, a, R, ncur, true);
    }
}

__device__ __forceinline__ double wsum_t(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// workgroup sum of a (re, im) pair broadcast to all threads; red >= 18 doubles
__device__ __forceinline__ double2 bsum(double2 v, double *red)
{
    v.x = wsum_t(v.x);
    v.y = wsum_t(v.y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (kTfftThreads + 63) >> 6;
    __syncthreads();
    if (lane == 0) { red[2 * wave] = v.x; red[2 * wave + 1] = v.y; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 t = make_double2(0.0, 0.0);
        for (int w = 0; w < nwave; ++w) { t.x += red[2 * w]; t.y += red[2 * w + 1]; }
        red[16] = t.x;
        red[17] = t.y;
    }
    __syncthreads();
    return make_double2(red[16], red[17]);
}

// two pairs at once (same shuffle tree and wave order per value as bsum: bit-identical sums); red >= 36 doubles
__device__ __forceinline__ void bsum4(double2 &u, double2 &v, double *red)
{
    u.x = wsum_t(u.x); u.y = wsum_t(u.y);
    v.x = wsum_t(v.x); v.y = wsum_t(v.y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (kTfftThreads + 63) >> 6;
    __syncthreads();
    if (lane == 0) { red[4 * wave] = u.x; red[4 * wave + 1] = u.y; red[4 * wave + 2] = v.x; red[4 * wave + 3] = v.y; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int w = 0; w < nwave; ++w)
            for (int q = 0; q < 4; ++q) t[q] += red[4 * w + q];
        for (int q = 0; q < 4; ++q) red[32 + q] = t[q];
    }
    __syncthreads();
    u = make_double2(red[32], red[33]);
    v = make_double2(red[34], red[35]);
}

__device__ __forceinline__ double2 reduce_c(const double2 *part, int n, double *red)
{
    double2 t = make_double2(0.0, 0.0);
    for (int c = threadIdx.x; c < n; c += kTfftThreads) { t.x += part[c].x; t.y += part[c].y; }
    return bsum(t, red);
}

__device__ __forceinline__ double reduce_r(const double *part, int n, double *red)
{
    double2 t = make_double2(0.0, 0.0);
    for (int c = threadIdx.x; c < n; c += kTfftThreads) t.x += part[c];
    return bsum(t, red).x;
}

__device__ __forceinline__ double2 cdivt(double2 a, double2 b)
{
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

enum { MODE_PLAIN_FWD = 0, MODE_PLAIN_INV = 1, MODE_FWD_CG = 2, MODE_INV_CG = 3 };

// SLIM: the in-place form (one LDS image, radix <= 7): about half the LDS and two thirds of the registers of the ping-pong form, so
// more workgroups fit a CU when several CG pipelines share the chip
typedef double v2d_t __attribute__((ext_vector_type(2)));
// ---- register-blocked form of the two-image transform (EDGE): Lτ = 4 · M · 4 with M = 4 or 8 and a tile of exactly 4 slices per lane
// (Lτ · SB = 1024).  A lane's four staged slices n1 + (Lτ/4)·n2 are the inputs of one radix-4 butterfly of a decimation-in-frequency first
// stage, and its four epilogue slices l0 + (Lτ/4)·u are the outputs of one radix-4 butterfly of a decimation-in-time last stage.  Both are
// done in registers, so the LDS sees ONE pass (radix M, in place: one image) instead of three (two-image form) or four (in-place form):
//   X[k2 + 4 k1] = Σ_n1 W_Lτ^{n1 k2} [Σ_n2 x[n1 + (Lτ/4) n2] (∓i)^{n2 k2}] W_{Lτ/4}^{n1 k1},     n1 = 4a + b,  k1 = k' + M u:
//   y_k2[n1] (staging)  ->  Z_{k2,b}[k'] = Σ_a y_k2[4a + b] W_M^{a k'} (LDS pass)  ->  X[l0 + (Lτ/4) u] = Σ_b (∓i)^{bu} W_Lτ^{4 b k'} Z_{k2,b}[k'],
// k2 = l0 mod 4, k' = l0 div 4.  Same arithmetic as the three Stockham passes up to the order of the additions.
template <int M>
__device__ __forceinline__ void edge_middle(double2 *X, int SB, bool inv)
{
    // work item = (b, k2, sb): the M values y_k2[4a + b] at X[((4a + b)·4 + k2)·SB + sb] -> Z_{k2,b}[k'] at X[((4k' + b)·4 + k2)·SB + sb]:
    // a butterfly writes the M rows it read, so the pass is in place and the form needs ONE LDS image
    for (int w = threadIdx.x; w < 16 * SB; w += kTfftThreads) {
        const int sb = w % SB, k2 = (w / SB) & 3, b = w / (4 * SB);
        double2 *x = X + (size_t)(b * 4 + k2) * SB + sb;
        double2 v[M];
#pragma unroll
        for (int a_ = 0; a_ < M; ++a_) v[a_] = x[(size_t)a_ * 16 * SB];
        dft<M>(v, nullptr, 0, inv);
#pragma unroll
        for (int kp = 0; kp < M; ++kp) x[(size_t)kp * 16 * SB] = v[kp];
    }
    __syncthreads();
}

template <int MODE, bool SLIM, bool EDGE = false>
__global__ void __launch_bounds__(256, (SLIM || EDGE) ? 6 : 1) tfft_kernel(TfftArgs a)
{
    static_assert(!(SLIM && EDGE), "the register-blocked form replaces both pass schedules");
    extern __shared__ double2 lds[];
    __shared__ double red[36];
    const int Lt = a.Lt, SB = a.SB, N = a.N;
    double2 *A = lds, *B = A + (size_t)Lt * SB, *WT = (SLIM || EDGE) ? B : B + (size_t)Lt * SB;  // one image in the in-place and register-blocked forms
    int *POS = reinterpret_cast<int *>(WT + Lt);  // SLIM only
    // XCD-aware order (workgroups go to the eight XCDs round-robin): XCD x works on the same contiguous share of the systems as in the MᵀM
    // and Chebyshev kernels — every kernel of the iteration then walks the same eighth of each vector on a given XCD (api_cg.hip,
    // cg_iteration_fused, for what that buys and what it does not)
    int bid_ = blockIdx.x;
    if (a.xcd_map && (gridDim.x & 7) == 0) bid_ = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tile = bid_ % a.ntile, sys = a.sys_first + bid_ / a.ntile;
    int st_done = 0, st_stop = 0;
    if (MODE == MODE_FWD_CG) {
        // `done` was written by an earlier launch (inverse kernel of the previous iteration or cg_start): safe to gate on.
        // Latch it into `stop` for the inverse kernel of THIS iteration, which must not look at `done` (it writes it).
        st_done = a.st[sys].done;  // acted on below, behind the first staging loads (one memory latency instead of two in a row)
    }
    // the recurrence scalars of this system, read with the stop flag (same cache line) instead of after the transform
    double st_normb2 = 1.0, st_tol = 0.0;
    double2 st_alpha = make_double2(0.0, 0.0), st_rho = make_double2(1.0, 0.0);
    if (MODE == MODE_INV_CG) {
        const CgState &s0 = a.st[sys];
        st_stop = s0.stop;         // likewise
        st_normb2 = s0.normb2; st_tol = s0.tol;
        st_alpha = make_double2(s0.alpha_re, s0.alpha_im);
        st_rho = make_double2(s0.rho_re, s0.rho_im);
    }
    const int i0 = tile * SB, ns = min(SB, N - i0);
    const size_t sstride = (size_t)a.nsys * N;
    const size_t base = (size_t)sys * N + i0;
    // One round of loads: the twiddle table and the row positions (first 256 entries: one per lane, into registers), the partial sums and
    // the tile's first batch of slices are all requested before any of them is waited for; the LDS copies of the tables are written
    // behind the staging loads (staging_tables below).  As first written (tables: load, wait, store; then the rest) every workgroup paid a
    // full memory round trip before it asked for its data.
    constexpr bool INV = (MODE == MODE_PLAIN_INV || MODE == MODE_INV_CG);
    const int tq_ = min((int)threadIdx.x, Lt - 1);
    const double2 wt0_ = a.wtab[tq_];
    int pos0_ = 0;
    if (SLIM) pos0_ = a.pos[tq_];
    auto staging_tables = [&]() {
        if ((int)threadIdx.x < Lt) {
            WT[threadIdx.x] = wt0_;
            if (SLIM) POS[threadIdx.x] = pos0_;
        }
        for (int q = threadIdx.x + kTfftThreads; q < Lt; q += kTfftThreads) {  // Lτ > 256 only
            WT[q] = a.wtab[q];
            if (SLIM) POS[q] = a.pos[q];
        }
        if (SLIM && INV) __syncthreads();  // the staging stores scatter through POS
    };

    // The scalars of the CG recurrence (α; the stop test and β) depend on the partial sums the previous kernels left, not on this
    // kernel's transform: their loads are issued here, in front of the staging loads, and reduced before the passes, so that their memory
    // latency and barriers are not a serial step between the transform and the epilogue.
    double2 q1 = make_double2(0.0, 0.0), q2 = make_double2(0.0, 0.0);
    if (MODE == MODE_FWD_CG) {
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride, *ppz = a.part_pz + (size_t)sys * a.pz_stride;
        // unconditional, clamped (see the staging loads): lanes beyond the count drop their value
        const double2 v1 = prz[min((int)threadIdx.x, a.nrz - 1)], v2 = ppz[min((int)threadIdx.x, a.npz - 1)];
        if ((int)threadIdx.x < a.nrz) q1 = v1;
        if ((int)threadIdx.x < a.npz) q2 = v2;
    } else if (MODE == MODE_INV_CG) {
        const double *prr = a.part_rr + (size_t)sys * a.rr_stride;
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride;
        const double v1 = prr[min((int)threadIdx.x, a.nrr - 1)];
        const double2 v2 = prz[min((int)threadIdx.x, a.nrz - 1)];
        if ((int)threadIdx.x < a.nrr) q1.x = v1;
        if ((int)threadIdx.x < a.nrz) q2 = v2;
    }

    // 256 % SB == 0, so a lane keeps one site column sb and walks slices l0, l0 + lstep, ...
    const int sb = threadIdx.x % SB, l0 = threadIdx.x / SB, lstep = kTfftThreads / SB;
    const bool act = sb < ns;
    constexpr int U = 4;  // slices in flight per lane: all loads of a batch are issued before its first store

    // staging: all loads of a batch of U slices are issued before the first LDS store — a plain load/store loop keeps ONE 16-byte load
    // per lane in flight, and with a single wave of workgroups on the chip the kernel is then bound by memory latency, not bandwidth
    {
        const double2 *src = (MODE == MODE_FWD_CG) ? a.z : a.src;  // forward CG mode transforms A p
        // one batch: loads (unconditional, clamped addresses — a load inside a branch makes the compiler lose count of what is in flight
        // and fall back to `s_waitcnt vmcnt(0)`, which here sat in front of the loads: a full round trip for the partial sums requested
        // above before the tile's data were even asked for), then the LDS stores
#define TFFT_STAGE_LOADS(l_)                                                                              \
        double2 t0_, t1_, t2_, t3_;                                                                       \
        {                                                                                                 \
            const size_t col_ = base + (act ? sb : 0);                                                    \
            t0_ = src[(size_t)min((l_), Lt - 1) * sstride + col_];                                        \
            t1_ = src[(size_t)min((l_) + lstep, Lt - 1) * sstride + col_];                                \
            t2_ = src[(size_t)min((l_) + 2 * lstep, Lt - 1) * sstride + col_];                            \
            t3_ = src[(size_t)min((l_) + 3 * lstep, Lt - 1) * sstride + col_];                            \
        }
#define TFFT_STAGE_STORE(t_, lu_)                                                                         \
        if ((lu_) < Lt) {                                                                                 \
            double2 x_ = act ? (t_) : make_double2(0.0, 0.0);                                             \
            if (MODE != MODE_FWD_CG && a.pre_tw && act) x_ = cm(x_, a.pre_tw[(lu_)]);                     \
            A[((SLIM && INV) ? POS[(lu_)] : (lu_)) * SB + sb] = x_;                                       \
        }
        static_assert(U == 4, "the staging macros are written for four slices in flight");
        {   // first batch, straight-line: the table copies and the early exits sit between its loads and its stores
            TFFT_STAGE_LOADS(l0)
            double2 ew1 = make_double2(1.0, 0.0), ew2 = ew1, ew3 = ew1;  // EDGE: W^{n1 k2}, n1 = l0
            if (EDGE) { ew1 = a.wtab[l0]; ew2 = a.wtab[2 * l0]; ew3 = a.wtab[3 * l0]; }
            staging_tables();
            // the early exits (workgroup-uniform), with the tile's first loads already in flight.  `done` was written by an earlier launch
            // (inverse kernel of the previous iteration or cg_start): safe to gate on; it is latched into `stop` for the inverse kernel of
            // THIS iteration, which must not look at `done` (it writes it).
            if (MODE == MODE_FWD_CG && st_done) {
                if (tile == 0 && threadIdx.x == 0) a.st[sys].stop = st_done;
                return;
            }
            if (MODE == MODE_INV_CG && st_stop) return;
            if (EDGE) {
                // first stage in registers: radix-4 butterfly over the lane's four slices, twiddles W^{n1 k2}, y_k2[n1] to A[(n1·4 + k2)·SB + sb]
                double2 v4[4] = {t0_, t1_, t2_, t3_};
                if (!act) v4[0] = v4[1] = v4[2] = v4[3] = make_double2(0.0, 0.0);
                if (MODE != MODE_FWD_CG && a.pre_tw && act) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v4[u] = cm(v4[u], a.pre_tw[l0 + u * lstep]);
                }
                dft<4>(v4, nullptr, 0, INV);
                if (INV) { ew1.y = -ew1.y; ew2.y = -ew2.y; ew3.y = -ew3.y; }
                A[(size_t)(l0 * 4 + 0) * SB + sb] = v4[0];
                A[(size_t)(l0 * 4 + 1) * SB + sb] = cm(v4[1], ew1);
                A[(size_t)(l0 * 4 + 2) * SB + sb] = cm(v4[2], ew2);
                A[(size_t)(l0 * 4 + 3) * SB + sb] = cm(v4[3], ew3);
            } else {
            TFFT_STAGE_STORE(t0_, l0)
            TFFT_STAGE_STORE(t1_, l0 + lstep)
            TFFT_STAGE_STORE(t2_, l0 + 2 * lstep)
            TFFT_STAGE_STORE(t3_, l0 + 3 * lstep)
            }
        }
        for (int l = l0 + U * lstep; l < Lt; l += U * lstep) {  // Lτ > U·lstep only
            TFFT_STAGE_LOADS(l)
            TFFT_STAGE_STORE(t0_, l)
            TFFT_STAGE_STORE(t1_, l + lstep)
            TFFT_STAGE_STORE(t2_, l + 2 * lstep)
            TFFT_STAGE_STORE(t3_, l + 3 * lstep)
        }
#undef TFFT_STAGE_LOADS
#undef TFFT_STAGE_STORE
        __syncthreads();
    }

    if (MODE == MODE_FWD_CG) {
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride, *ppz = a.part_pz + (size_t)sys * a.pz_stride;
        for (int c = threadIdx.x + kTfftThreads; c < a.nrz; c += kTfftThreads) { q1.x += prz[c].x; q1.y += prz[c].y; }
        for (int c = threadIdx.x + kTfftThreads; c < a.npz; c += kTfftThreads) { q2.x += ppz[c].x; q2.y += ppz[c].y; }
        bsum4(q1, q2, red);  // q1 = r·z, q2 = p·Ap
    } else if (MODE == MODE_INV_CG) {
        const double *prr = a.part_rr + (size_t)sys * a.rr_stride;
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride;
        for (int c = threadIdx.x + kTfftThreads; c < a.nrr; c += kTfftThreads) q1.x += prr[c];
        for (int c = threadIdx.x + kTfftThreads; c < a.nrz; c += kTfftThreads) { q2.x += prz[c].x; q2.y += prz[c].y; }
        bsum4(q1, q2, red);  // q1.x = |r|², q2 = r·z
    }

    const double2 *res = A;
    double2 eo[4];  // EDGE: the lane's four outputs X[l0 + lstep·u]
    if (EDGE) {
        if (a.edge == 8) edge_middle<8>(A, SB, INV);
        else edge_middle<4>(A, SB, INV);
        // last stage in registers: Z_{k2,b}[k'] for b = 0..3, twiddles W^{4 b k'}, radix-4 butterfly
        const int k2 = l0 & 3, kp = l0 >> 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double2 z = A[(size_t)((kp * 4 + b) * 4 + k2) * SB + sb];
            if (b > 0) {
                double2 w = WT[4 * b * kp];
                if (INV) w.y = -w.y;
                z = cm(z, w);
            }
            eo[b] = z;
        }
        dft<4>(eo, nullptr, 0, INV);
    } else if (SLIM) {
        if (INV) inplace_inverse(A, WT, a);
        else inplace_forward(A, WT, a);
        res = A;
    } else {
        res = stockham(A, B, WT, a, INV);
    }
    // LDS row of output element l
    auto row = [&](int l) { return (SLIM && !INV) ? POS[l] : l; };
    (void)eo;

    if (MODE == MODE_FWD_CG) {
        // ConjugateGradient.jl:219-226 in frequency space: α = (r·z)/(p·Ap), r̂ -= α·FFT(Ap), |r|² = Σ|r̂|²/Lτ
        const double2 rz = q1, pz = q2;
        const double2 alpha = cdivt(rz, pz);
        double acc = 0.0;
        if (act) {
            for (int l = l0; l < Lt; l += U * lstep) {
                double2 rv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int lu = l + u * lstep;
                    if (lu < Lt) rv[u] = a.r[(size_t)lu * sstride + base + sb];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int lu = l + u * lstep;
                    if (lu < Lt) {
                        const double2 rn = csub(rv[u], cm(alpha, EDGE ? eo[u] : res[row(lu) * SB + sb]));
                        a.r[(size_t)lu * sstride + base + sb] = rn;
                        acc += rn.x * rn.x + rn.y * rn.y;
                    }
                }
            }
        }
        const double2 t = bsum(make_double2(acc, 0.0), red);
        if (threadIdx.x == 0) {
            a.part_rr[(size_t)sys * a.rr_stride + tile] = t.x / Lt;
            if (tile == 0) {
                CgState &s = a.st[sys];
                s.rho_re = rz.x; s.rho_im = rz.y;
                s.alpha_re = alpha.x; s.alpha_im = alpha.y;
            }
        }
    } else if (MODE == MODE_INV_CG) {
        // ConjugateGradient.jl:220 (x += α p, deferred to here), :229-245: stop test on the unpreconditioned residual, p = z + β p
        const double rr = q1.x;
        const double eps = sqrt(rr) / sqrt(st_normb2);
        const bool conv = eps < st_tol;
        const double2 alpha = st_alpha;
        double2 beta = make_double2(0.0, 0.0);
        if (!conv) beta = cdivt(q2, st_rho);
        if (act) {
            for (int l = l0; l < Lt; l += U * lstep) {
                double2 pv[U], xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int lu = l + u * lstep;
                    if (lu < Lt) {
                        const size_t off = (size_t)lu * sstride + base + sb;
                        pv[u] = a.p[off];
                        if (a.x_stream) {  // x is touched by this kernel only, once per iteration: keep it out of the caches the other vectors live in
                            const v2d_t t_ = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(a.x + off));
                            xv[u] = make_double2(t_.x, t_.y);
                        } else xv[u] = a.x[off];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int lu = l + u * lstep;
                    if (lu < Lt) {
                        const size_t off = (size_t)lu * sstride + base + sb;
                        const double2 xn_ = cadd(xv[u], cm(alpha, pv[u]));
                        if (a.x_stream) { v2d_t t_; t_.x = xn_.x; t_.y = xn_.y; __builtin_nontemporal_store(t_, reinterpret_cast<v2d_t *>(a.x + off)); }
                        else a.x[off] = xn_;
                        if (!conv) a.p[off] = cadd(EDGE ? eo[u] : res[lu * SB + sb], cm(beta, pv[u]));
                    }
                }
            }
        }
        if (threadIdx.x == 0 && tile == 0) {
            CgState &s = a.st[sys];
            s.eps = eps;
            s.iters += 1;
            if (conv) s.done = 1;
            else if (s.iters >= s.maxiter) s.done = 2;
        }
    } else if (EDGE) {
        // plain modes: the lane's four outputs sit in registers (same (slice, site) map as the loop below: idx = l·SB + sb, l = l0 + lstep·u)
        if (act) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = l0 + u * lstep;
                double2 x = eo[u];
                if (a.post_tw) { const double2 w = a.post_tw[l]; x = cm(x, make_double2(w.x, -w.y)); }
                a.dst[(size_t)l * sstride + base + sb] = x;
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < Lt * SB; idx += kTfftThreads) {
            const int l = idx / SB, sb = idx - l * SB;
            if (sb < ns) {
                double2 x = res[row(l) * SB + sb];
                if (a.post_tw) { const double2 w = a.post_tw[l]; x = cm(x, make_double2(w.x, -w.y)); }
                a.dst[(size_t)l * sstride + base + sb] = x;
            }
        }
    }
}


// ---- register-blocked form for the lengths Lτ = R · M · R that are not 4 · M · 4 on 256 lanes: Lτ = 80 (4·5·4), 100 (5·4·5), 200 (5·8·5) —
// the time extents of BASELINE configs 2, 3 and 5.  R · M · SB lanes (320 for all three with the plan's tile width), R slices per lane:
//   X[k2 + R k1] = Σ_n1 W_Lτ^{n1 k2} [Σ_n2 x[n1 + (Lτ/R) n2] W_R^{n2 k2}] W_{Lτ/R}^{n1 k1},      n1 = R a + b,  k1 = k' + M u:
//   y_k2[n1] (radix-R butterfly on the lane's R staged slices n1 + (Lτ/R)·n2, in registers)
//     ->  Z_{k2,b}[k'] = Σ_a y_k2[R a + b] W_M^{a k'}   (the ONE LDS pass, in place)
//     ->  X[l0 + (Lτ/R) u] = Σ_b W_R^{b u} W_Lτ^{R b k'} Z_{k2,b}[k']   (radix-R butterfly in registers: the lane's R epilogue slices),
// k2 = l0 mod R, k' = l0 div R.  Same structure as tfft_kernel<…, EDGE> (R = 4, 256 lanes), which keeps Lτ = 64 and 128 bit for bit;
// here the mixed-radix lengths go from three or four LDS passes to one (the chain's τ-FFTs were 55 % of its CG iteration).
template <int NT>
__device__ __forceinline__ void bsum4_n(double2 &u, double2 &v, double *red)
{
    u.x = wsum_t(u.x); u.y = wsum_t(u.y);
    v.x = wsum_t(v.x); v.y = wsum_t(v.y);
    constexpr int nwave = (NT + 63) >> 6;
    static_assert(nwave <= 8, "red[] holds eight wave sums of four values");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[4 * wave] = u.x; red[4 * wave + 1] = u.y; red[4 * wave + 2] = v.x; red[4 * wave + 3] = v.y; }
    __syncthreads();
    double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int w = 0; w < nwave; ++w)  // every lane adds the wave sums in the same fixed order: no second barrier, no broadcast slot
#pragma unroll
        for (int q = 0; q < 4; ++q) t[q] += red[4 * w + q];
    u = make_double2(t[0], t[1]);
    v = make_double2(t[2], t[3]);
}

template <int MODE, int R, int M, int SB>
__global__ void __launch_bounds__(R * M * SB, 4) tfft_rb_kernel(TfftArgs a)
{
    constexpr int NT = R * M * SB, LT = R * M * R, LSTEP = M * R;
    constexpr bool INV = (MODE == MODE_PLAIN_INV || MODE == MODE_INV_CG);
    static_assert(NT % 64 == 0 && NT <= 512, "whole wavefronts");
    extern __shared__ double2 lds[];
    __shared__ double red[40];   // [0, 32): wave sums of bsum4_n; [32, 40): wave sums of the epilogue's |r|²
    double2 *A = lds, *WT = A + (size_t)LT * SB;
    const int N = a.N;
    int bid_ = blockIdx.x;
    if (a.xcd_map && (gridDim.x & 7) == 0) bid_ = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // as in tfft_kernel
    const int tile = bid_ % a.ntile, sys = a.sys_first + bid_ / a.ntile;
    int st_done = 0, st_stop = 0;
    if (MODE == MODE_FWD_CG) st_done = a.st[sys].done;
    double st_normb2 = 1.0, st_tol = 0.0;
    double2 st_alpha = make_double2(0.0, 0.0), st_rho = make_double2(1.0, 0.0);
    if (MODE == MODE_INV_CG) {
        const CgState &s0 = a.st[sys];
        st_stop = s0.stop;
        st_normb2 = s0.normb2; st_tol = s0.tol;
        st_alpha = make_double2(s0.alpha_re, s0.alpha_im);
        st_rho = make_double2(s0.rho_re, s0.rho_im);
    }
    const int i0 = tile * SB, ns = min(SB, N - i0);
    const size_t sstride = (size_t)a.nsys * N;
    const size_t base = (size_t)sys * N + i0;
    const int sb = threadIdx.x % SB, l0 = threadIdx.x / SB;  // l0 in [0, M·R): the lane's slices are l0 + LSTEP·u, u < R
    const bool act = sb < ns;
    const size_t col = base + (act ? sb : 0);

    // one round of loads, nothing waited for in between: twiddle table entry, partial sums of the previous kernels, the lane's R slices,
    // the first-stage twiddles
    const double2 wt0_ = a.wtab[min((int)threadIdx.x, LT - 1)];
    double2 q1 = make_double2(0.0, 0.0), q2 = make_double2(0.0, 0.0);
    if (MODE == MODE_FWD_CG) {
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride, *ppz = a.part_pz + (size_t)sys * a.pz_stride;
        const double2 v1 = prz[min((int)threadIdx.x, a.nrz - 1)], v2 = ppz[min((int)threadIdx.x, a.npz - 1)];
        if ((int)threadIdx.x < a.nrz) q1 = v1;
        if ((int)threadIdx.x < a.npz) q2 = v2;
    } else if (MODE == MODE_INV_CG) {
        const double *prr = a.part_rr + (size_t)sys * a.rr_stride;
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride;
        const double v1 = prr[min((int)threadIdx.x, a.nrr - 1)];
        const double2 v2 = prz[min((int)threadIdx.x, a.nrz - 1)];
        if ((int)threadIdx.x < a.nrr) q1.x = v1;
        if ((int)threadIdx.x < a.nrz) q2 = v2;
    }
    const double2 *src = (MODE == MODE_FWD_CG) ? a.z : a.src;  // forward CG mode transforms A p
    double2 v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = src[(size_t)(l0 + u * LSTEP) * sstride + col];
    double2 ew[R];
#pragma unroll
    for (int k2 = 1; k2 < R; ++k2) ew[k2] = a.wtab[k2 * l0];  // W_Lτ^{n1 k2}, n1 = l0 < Lτ/R
    if ((int)threadIdx.x < LT) WT[threadIdx.x] = wt0_;
    if (NT < LT) for (int q = threadIdx.x + NT; q < LT; q += NT) WT[q] = a.wtab[q];
    // the early exits (workgroup-uniform), with the loads above in flight — see tfft_kernel for the done / stop protocol
    if (MODE == MODE_FWD_CG && st_done) {
        if (tile == 0 && threadIdx.x == 0) a.st[sys].stop = st_done;
        return;
    }
    if (MODE == MODE_INV_CG && st_stop) return;
    // first stage in registers
    if (!act) {
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = make_double2(0.0, 0.0);
    }
    if (MODE != MODE_FWD_CG && MODE != MODE_INV_CG && a.pre_tw && act) {
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = cm(v[u], a.pre_tw[l0 + u * LSTEP]);
    }
    dft<R>(v, nullptr, 0, INV);
    A[(size_t)(l0 * R) * SB + sb] = v[0];
#pragma unroll
    for (int k2 = 1; k2 < R; ++k2) {
        double2 w = ew[k2];
        if (INV) w.y = -w.y;
        A[(size_t)(l0 * R + k2) * SB + sb] = cm(v[k2], w);
    }
    // the scalars of the recurrence: remaining partial sums, wave sums into red[] — the barrier that publishes them is the one the
    // LDS image needs anyway
    if (MODE == MODE_FWD_CG) {
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride, *ppz = a.part_pz + (size_t)sys * a.pz_stride;
        for (int c = threadIdx.x + NT; c < a.nrz; c += NT) { q1.x += prz[c].x; q1.y += prz[c].y; }
        for (int c = threadIdx.x + NT; c < a.npz; c += NT) { q2.x += ppz[c].x; q2.y += ppz[c].y; }
        bsum4_n<NT>(q1, q2, red);  // q1 = r·z, q2 = p·Ap
    } else if (MODE == MODE_INV_CG) {
        const double *prr = a.part_rr + (size_t)sys * a.rr_stride;
        const double2 *prz = a.part_rz + (size_t)sys * a.rz_stride;
        for (int c = threadIdx.x + NT; c < a.nrr; c += NT) q1.x += prr[c];
        for (int c = threadIdx.x + NT; c < a.nrz; c += NT) { q2.x += prz[c].x; q2.y += prz[c].y; }
        bsum4_n<NT>(q1, q2, red);  // q1.x = |r|², q2 = r·z
    } else {
        __syncthreads();
    }
    // the LDS pass: work item (b, k2, sb) = w holds the M values y_k2[R a + b] at A[w + a·R²·SB] -> Z_{k2,b}[k'] at A[w + k'·R²·SB]
    for (int w = threadIdx.x; w < R * R * SB; w += NT) {
        double2 *x = A + w;
        double2 m_[M];
#pragma unroll
        for (int a_ = 0; a_ < M; ++a_) m_[a_] = x[(size_t)a_ * R * R * SB];
        dft<M>(m_, nullptr, 0, INV);
#pragma unroll
        for (int kp = 0; kp < M; ++kp) x[(size_t)kp * R * R * SB] = m_[kp];
    }
    __syncthreads();
    // last stage in registers: the lane's outputs are the slices l0 + LSTEP·u
    double2 eo[R];
    {
        const int k2 = l0 % R, kp = l0 / R;
#pragma unroll
        for (int b = 0; b < R; ++b) {
            double2 z = A[(size_t)((kp * R + b) * R + k2) * SB + sb];
            if (b > 0) {
                double2 w = WT[R * b * kp];
                if (INV) w.y = -w.y;
                z = cm(z, w);
            }
            eo[b] = z;
        }
        dft<R>(eo, nullptr, 0, INV);
    }

    if (MODE == MODE_FWD_CG) {
        // ConjugateGradient.jl:219-226 in frequency space: α = (r·z)/(p·Ap), r̂ -= α·FFT(Ap), |r|² = Σ|r̂|²/Lτ
        const double2 rz = q1, pz = q2;
        const double2 alpha = cdivt(rz, pz);
        double acc = 0.0;
        if (act) {
            double2 rv[R];
#pragma unroll
            for (int u = 0; u < R; ++u) rv[u] = a.r[(size_t)(l0 + u * LSTEP) * sstride + base + sb];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const double2 rn = csub(rv[u], cm(alpha, eo[u]));
                a.r[(size_t)(l0 + u * LSTEP) * sstride + base + sb] = rn;
                acc += rn.x * rn.x + rn.y * rn.y;
            }
        }
        acc = wsum_t(acc);
        if ((threadIdx.x & 63) == 0) red[32 + (threadIdx.x >> 6)] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) t += red[32 + w];
            a.part_rr[(size_t)sys * a.rr_stride + tile] = t / LT;
            if (tile == 0) {
                CgState &s = a.st[sys];
                s.rho_re = rz.x; s.rho_im = rz.y;
                s.alpha_re = alpha.x; s.alpha_im = alpha.y;
            }
        }
    } else if (MODE == MODE_INV_CG) {
        // ConjugateGradient.jl:220 (x += α p, deferred to here), :229-245: stop test on the unpreconditioned residual, p = z + β p
        const double rr = q1.x;
        const double eps = sqrt(rr) / sqrt(st_normb2);
        const bool conv = eps < st_tol;
        const double2 alpha = st_alpha;
        double2 beta = make_double2(0.0, 0.0);
        if (!conv) beta = cdivt(q2, st_rho);
        if (act) {
            double2 pv[R], xv[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const size_t off = (size_t)(l0 + u * LSTEP) * sstride + base + sb;
                pv[u] = a.p[off];
                if (a.x_stream) {
                    const v2d_t t_ = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(a.x + off));
                    xv[u] = make_double2(t_.x, t_.y);
                } else xv[u] = a.x[off];
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const size_t off = (size_t)(l0 + u * LSTEP) * sstride + base + sb;
                const double2 xn_ = cadd(xv[u], cm(alpha, pv[u]));
                if (a.x_stream) { v2d_t t_; t_.x = xn_.x; t_.y = xn_.y; __builtin_nontemporal_store(t_, reinterpret_cast<v2d_t *>(a.x + off)); }
                else a.x[off] = xn_;
                if (!conv) a.p[off] = cadd(eo[u], cm(beta, pv[u]));
            }
        }
        if (threadIdx.x == 0 && tile == 0) {
            CgState &s = a.st[sys];
            s.eps = eps;
            s.iters += 1;
            if (conv) s.done = 1;
            else if (s.iters >= s.maxiter) s.done = 2;
        }
    } else {
        if (act) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int l = l0 + u * LSTEP;
                double2 x = eo[u];
                if (a.post_tw) { const double2 w = a.post_tw[l]; x = cm(x, make_double2(w.x, -w.y)); }
                a.dst[(size_t)l * sstride + base + sb] = x;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// EFA leapfrog kernel.  Workgroup = (tile of SB phonon modes, walker), every τ of the tile in LDS, z = x + i p transformed with the
// same Stockham passes as the state vectors (one complex transform carries both real fields: x̃_ω = (z̃_ω + conj z̃_{-ω})/2,
// p̃_ω = (z̃_ω − conj z̃_{-ω})/(2i)).  Per mode the harmonic flow of H = |p̃|²/(2m) + q|x̃|²/2 is exact:
//   x̃(t) = cos(wt) x̃ + sin(wt)/(m w) p̃,   p̃(t) = cos(wt) p̃ − m w sin(wt) x̃,   w = √(q/m)
// (w = 1 for every mode when m = q: "exact Fourier acceleration").  Energies use the unitary normalisation |·|²/Lτ.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) efa_kernel(EfaArgs a, TfftArgs plan)
{
    extern __shared__ double2 lds[];
    __shared__ double red[18];
    const int Lt = a.Lt, SB = a.SB, Nph = a.Nph;
    double2 *A = lds, *B = A + (size_t)Lt * SB, *WT = B + (size_t)Lt * SB;
    const int tile = blockIdx.x % a.ntile, w = blockIdx.x / a.ntile;
    const int i0 = tile * SB, ns = min(SB, Nph - i0);
    const size_t wbase = (size_t)w * Lt * Nph;
    for (int q = threadIdx.x; q < Lt; q += kTfftThreads) WT[q] = a.wtab[q];
    for (int idx = threadIdx.x; idx < Lt * SB; idx += kTfftThreads) {
        const int l = idx / SB, sb = idx - l * SB;
        double2 z = make_double2(0.0, 0.0);
        if (sb < ns) {
            const size_t off = wbase + (size_t)l * Nph + i0 + sb;
            z = make_double2(a.mode == 1 ? 0.0 : a.x[off], a.p[off]);
            if (a.force) z.y -= a.kick * a.force[off];
            if (a.mode == 1) z = make_double2(z.y, 0.0);  // transform the noise as a real field
        }
        A[idx] = z;
    }
    __syncthreads();
    double2 *res = stockham(A, B, WT, plan, false);
    double2 *oth = (res == A) ? B : A;
    double accK = 0.0, accS = 0.0;
    const double iLt = 1.0 / (double)Lt;
    for (int idx = threadIdx.x; idx < Lt * SB; idx += kTfftThreads) {
        const int om = idx / SB, sb = idx - om * SB;
        double2 out = make_double2(0.0, 0.0);
        if (sb < ns) {
            const int omm = om == 0 ? 0 : Lt - om;
            const double2 zp = res[idx], zm = res[omm * SB + sb];
            double2 X = make_double2(0.5 * (zp.x + zm.x), 0.5 * (zp.y - zm.y));   // x̃_ω
            double2 P = make_double2(0.5 * (zp.y + zm.y), -0.5 * (zp.x - zm.x));  // p̃_ω = (z̃_ω − conj z̃_{-ω})/(2i)
            const double q = a.q[(size_t)om * Nph + i0 + sb], m = a.m[(size_t)om * Nph + i0 + sb];
            const bool live = a.finite_mass[i0 + sb] != 0 && m > 0.0 && isfinite(m);
            if (a.mode == 1) {
                // p̃ = √m R̃ (covariance F⁻¹ diag(m) F); a frozen mode gets no momentum
                const double f = live ? sqrt(m) : 0.0;
                out = make_double2(f * zp.x, f * zp.y);
                P = out; X = make_double2(0.0, 0.0);
            } else if (a.mode == 0 && live) {
                const double wq = sqrt(q / m);
                double sn, cs;
                sincos(wq * a.dt, &sn, &cs);
                const double f1 = wq > 0.0 ? sn / (m * wq) : a.dt / m, f2 = m * wq * sn;
                const double2 Xn = make_double2(cs * X.x + f1 * P.x, cs * X.y + f1 * P.y);
                const double2 Pn = make_double2(cs * P.x - f2 * X.x, cs * P.y - f2 * X.y);
                X = Xn; P = Pn;
                out = make_double2(X.x - P.y, X.y + P.x);  // z̃' = x̃' + i p̃'
            } else {
                out = zp;
            }
            if (live) {
                accK += 0.5 * (P.x * P.x + P.y * P.y) / m * iLt;
                accS += 0.5 * q * (X.x * X.x + X.y * X.y) * iLt;
            }
        }
        oth[idx] = out;
    }
    __syncthreads();
    if (a.mode != 2) {
        res = stockham(oth, res, WT, plan, true);
        for (int idx = threadIdx.x; idx < Lt * SB; idx += kTfftThreads) {
            const int l = idx / SB, sb = idx - l * SB;
            if (sb < ns && (a.mode == 1 || a.finite_mass[i0 + sb] != 0)) {  // an infinite-mass mode is not touched at all (its momentum is set to zero when sampled)
                const size_t off = wbase + (size_t)l * Nph + i0 + sb;
                const double2 z = res[idx];
                if (a.mode == 0) { a.x[off] = z.x * iLt; a.p[off] = z.y * iLt; }
                else a.p[off] = z.x * iLt;  // real by the ω ↔ −ω symmetry of m
            }
        }
    }
    if (a.part) {
        const double2 t = bsum(make_double2(accK, accS), red);
        if (threadIdx.x == 0) { a.part[2 * (size_t)blockIdx.x] = t.x; a.part[2 * (size_t)blockIdx.x + 1] = t.y; }
    }
}

}  // namespace

bool tfft_plan(int Lt, int N, TfftArgs &a)
{
    a.Lt = Lt; a.N = N; a.nfac = 0;
    int m = Lt;
    const int pref[] = {8, 4, 2, 5, 3, 7};
    for (int R : pref)
        while (m % R == 0 && a.nfac < 16) { a.fac[a.nfac++] = R; m /= R; }
    if (m != 1) return false;
    // fewer, larger passes: a trailing (8, 2) pair reads better as (4, 4)
    for (int f = 0; f + 1 < a.nfac; ++f)
        if (a.fac[f] == 8 && a.fac[f + 1] == 2) { a.fac[f] = 4; a.fac[f + 1] = 4; }
    // in-place plan: radix 4, 2, 3, 5
    a.snfac = 0;
    m = Lt;
    const int spref[] = {4, 2, 3, 5};
    for (int R : spref)
        while (m % R == 0 && a.snfac < 16) { a.sfac[a.snfac++] = R; m /= R; }
    // the kernels read the radices from these 4-bit packs (scalar shifts), not from the arrays (a global load and a wait per pass)
    a.fpack = a.sfpack = 0;
    for (int f = 0; f < a.nfac; ++f) a.fpack |= (unsigned long long)a.fac[f] << (4 * f);
    for (int f = 0; f < a.snfac; ++f) a.sfpack |= (unsigned long long)a.sfac[f] << (4 * f);
    static const int slim_env = tuning_env(kTuneTfftSlim) > 0 ? tuning_env(kTuneTfftSlim) : 0;
    a.x_stream = 0;  // decided per launch (api_cg.hip: cg_iteration_fused)
    a.xcd_map = 0;   // decided per launch (api_cg.hip: cg_iteration_fused)
    a.slim_ok = (m == 1) ? 1 : 0;           // lengths 2^a 3^b 5^c only
    a.slim = (slim_env && a.slim_ok) ? 1 : 0;  // default form: SMOQY_TFFT_SLIM, else smoqy_tfft_form
    a.SB = 16;
    size_t lds_cap = 64 * 1024;  // two or more workgroups per CU
    {   // tuning knob SMOQY_TFFT_SB: sites per tile (4, 8 or 16), also lifts the LDS cap
        const int v = tuning_env(kTuneTfftSb);
        if (v == 4 || v == 8 || v == 16) { a.SB = v; lds_cap = 150 * 1024; }
    }
    // the tile width follows the ping-pong footprint in both forms (the EFA kernel always runs the ping-pong passes)
    while (a.SB > 4 && (2 * (size_t)Lt * a.SB + Lt) * sizeof(double2) > lds_cap) a.SB /= 2;
    if ((2 * (size_t)Lt * a.SB + Lt) * sizeof(double2) > 150 * 1024) return false;
    a.ntile = (N + a.SB - 1) / a.SB;
    // register-blocked two-image form (tfft_kernel<…, EDGE>): Lτ = 64 or 128 with exactly four slices per lane; SMOQY_TFFT_EDGE=0 switches it off
    static const int edge_env = tuning_env(kTuneTfftEdge) == 0 ? 0 : 1;
    a.edge = (edge_env && (Lt == 64 || Lt == 128) && Lt * a.SB == 4 * kTfftThreads) ? Lt / 16 : 0;
    // tfft_rb_kernel: Lτ = 80, 100, 200 at the tile width chosen above (320 lanes); SMOQY_TFFT_EDGE=1 keeps only the 256-lane form above
    static const int rb_env = (tuning_env(kTuneTfftEdge) < 0 || tuning_env(kTuneTfftEdge) >= 2) ? 1 : 0;
    a.rb = 0;
    if (rb_env) {
        if (Lt == 80 && a.SB == 16) a.rb = 16 * 4 + 5;
        else if (Lt == 100 && a.SB == 16) a.rb = 16 * 5 + 4;
        else if (Lt == 200 && a.SB == 8) a.rb = 16 * 5 + 8;
    }
    return true;
}

// row of spectrum element k after the decimation-in-frequency passes sfac[0], sfac[1], ...: the pass of radix R on a block of n rows
// sends the elements k ≡ s (mod R) to sub-block s (rows s·n/R ...), where the rest of the passes sort k div R
void tfft_positions(const TfftArgs &a, int *pos)
{
    for (int k = 0; k < a.Lt; ++k) {
        int n = a.Lt, kk = k, p = 0;
        for (int f = 0; f < a.snfac; ++f) {
            const int R = a.sfac[f];
            n /= R;
            p += (kk % R) * n;
            kk /= R;
        }
        pos[k] = p;
    }
}

hipError_t configure_tfft_kernels(const char **what)
{
    hipError_t first = hipSuccess;
    SMOQY_SET_LDS((tfft_kernel<0, false>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<1, false>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<2, false>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<3, false>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<0, false, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<1, false, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<2, false, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<3, false, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<0, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<1, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<2, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS((tfft_kernel<3, true>), 160 * 1024 - 512);
    SMOQY_SET_LDS(efa_kernel, 160 * 1024 - 512);
    return first;  // tfft_rb_kernel stays under the default 64 KB of dynamic LDS (29 KB at most)
}

void launch_tfft(hipStream_t st, int mode, const TfftArgs &a)
{
    const dim3 grid((unsigned)(a.ntile * (a.sys_count > 0 ? a.sys_count : a.nsys))), block(kTfftThreads);
    if (a.edge) {  // register-blocked form (Lτ = 64, 128): one image, one pass — stands in for both the two-image and the in-place request
        const size_t lds = ((size_t)a.Lt * a.SB + a.Lt) * sizeof(double2);
        switch (mode) {
            case 0: hipLaunchKernelGGL((tfft_kernel<0, false, true>), grid, block, lds, st, a); break;
            case 1: hipLaunchKernelGGL((tfft_kernel<1, false, true>), grid, block, lds, st, a); break;
            case 2: hipLaunchKernelGGL((tfft_kernel<2, false, true>), grid, block, lds, st, a); break;
            default: hipLaunchKernelGGL((tfft_kernel<3, false, true>), grid, block, lds, st, a); break;
        }
        return;
    }
    if (a.rb) {  // register-blocked form for Lτ = R·M·R (80, 100, 200): one image, one pass — likewise stands in for both requests
        const size_t lds = ((size_t)a.Lt * a.SB + a.Lt) * sizeof(double2);
#define SMOQY_RB_LAUNCH(R_, M_, SB_)                                                                                        \
        {                                                                                                                   \
            const dim3 blk((R_) * (M_) * (SB_));                                                                            \
            switch (mode) {                                                                                                 \
                case 0: hipLaunchKernelGGL((tfft_rb_kernel<0, R_, M_, SB_>), grid, blk, lds, st, a); break;                 \
                case 1: hipLaunchKernelGGL((tfft_rb_kernel<1, R_, M_, SB_>), grid, blk, lds, st, a); break;                 \
                case 2: hipLaunchKernelGGL((tfft_rb_kernel<2, R_, M_, SB_>), grid, blk, lds, st, a); break;                 \
                default: hipLaunchKernelGGL((tfft_rb_kernel<3, R_, M_, SB_>), grid, blk, lds, st, a); break;                \
            }                                                                                                               \
        }
        if (a.rb == 16 * 4 + 5) SMOQY_RB_LAUNCH(4, 5, 16)
        else if (a.rb == 16 * 5 + 4) SMOQY_RB_LAUNCH(5, 4, 16)
        else SMOQY_RB_LAUNCH(5, 8, 8)
#undef SMOQY_RB_LAUNCH
        return;
    }
    if (a.slim && a.pos) {
        const size_t lds = ((size_t)a.Lt * a.SB + a.Lt) * sizeof(double2) + (size_t)a.Lt * sizeof(int);
        switch (mode) {
            case 0: hipLaunchKernelGGL((tfft_kernel<0, true>), grid, block, lds, st, a); break;
            case 1: hipLaunchKernelGGL((tfft_kernel<1, true>), grid, block, lds, st, a); break;
            case 2: hipLaunchKernelGGL((tfft_kernel<2, true>), grid, block, lds, st, a); break;
            default: hipLaunchKernelGGL((tfft_kernel<3, true>), grid, block, lds, st, a); break;
        }
        return;
    }
    const size_t lds = (2 * (size_t)a.Lt * a.SB + a.Lt) * sizeof(double2);
    switch (mode) {
        case 0: hipLaunchKernelGGL((tfft_kernel<0, false>), grid, block, lds, st, a); break;
        case 1: hipLaunchKernelGGL((tfft_kernel<1, false>), grid, block, lds, st, a); break;
        case 2: hipLaunchKernelGGL((tfft_kernel<2, false>), grid, block, lds, st, a); break;
        default: hipLaunchKernelGGL((tfft_kernel<3, false>), grid, block, lds, st, a); break;
    }
}

void launch_efa(hipStream_t st, const EfaArgs &a)
{
    TfftArgs plan{};
    plan.Lt = a.Lt; plan.N = a.Nph; plan.nsys = a.nw; plan.SB = a.SB; plan.ntile = a.ntile; plan.nfac = a.nfac;
    for (int f = 0; f < 16; ++f) plan.fac[f] = a.fac[f];
    plan.fpack = 0;
    for (int f = 0; f < a.nfac; ++f) plan.fpack |= (unsigned long long)a.fac[f] << (4 * f);
    plan.wtab = a.wtab;
    const size_t lds = (2 * (size_t)a.Lt * a.SB + a.Lt) * sizeof(double2);
    hipLaunchKernelGGL(efa_kernel, dim3((unsigned)(a.ntile * a.nw)), dim3(kTfftThreads), lds, st, a, plan);
}

}  // namespace smoqy
