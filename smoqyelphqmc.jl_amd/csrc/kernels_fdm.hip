// FermionDetMatrix applies for gfx950: M, Mᵀ, MᵀM and MMᵀ, each as ONE launch.
//
// Reference semantics: src/FermionDetMatrix.jl:385-427 / 430-466 (mul_M!), :484-525 / 528-563
// (mul_Mt!), :329-340 (mul_MtM!), :357-368 (mul_MMt!), with the bond factors of
// src/checkerboard_matrix_multiply.jl:50-69.
//
// Mapping: the only coupling along imaginary time is the one-slice shift, so a
// workgroup owns a chunk of Tc consecutive time slices of one system and ALL N sites of those
// slices.  The slices live in LDS as contiguous N-vectors; every checkerboard colour is one
// barrier-separated stage in which each lane owns one bond (two disjoint sites) for all slices
// of the chunk.  The per-bond cosh/sinh pairs are read with lane == bond (coalesced).  MᵀM is
// fused by recomputing one halo slice: (M p) is formed for Tc+1 slices in LDS, then Mᵀ is applied
// to it without ever writing M p to memory.  An optional per-workgroup partial of dot(in, out)
// (wavefront shuffles + one LDS hop) feeds the CG's p·Ap without another pass.
#include "smoqy_internal.h"

namespace smoqy {

// v - ph·u, see hopcomb in kernels_fdm_fast.hip
__device__ __forceinline__ double2 hopcomb_g(double2 v, double2 u, bool wrapped, bool dagger, const FdmArgs &a)
{
    double pr = a.hop_re, pi = dagger ? -a.hop_im : a.hop_im;
    if (wrapped && a.antiperiodic) { pr = -pr; pi = -pi; }
    return make_double2(v.x - (pr * u.x - pi * u.y), v.y - (pr * u.y + pi * u.x));
}

__device__ __forceinline__ int wrap(int l, int Lt) { return l >= Lt ? l - Lt : (l < 0 ? l + Lt : l); }

// One checkerboard colour on nk LDS-resident slices: lane = bond.
// shi != nullptr: complex hopping, factor [[c, s], [conj(s), c]] with s = sh + i shi (checkerboard_matrix_multiply.jl:60-68)
__device__ __forceinline__ void colour_stage(double2 *U, int nk, int N, int lbase, int Lt, int Nh, const int2 *__restrict__ bonds, const double *__restrict__ ch,
                                             const double *__restrict__ sh, const double *__restrict__ shi, int cb, int ce)
{
    for (int h = cb + (int)threadIdx.x; h < ce; h += (int)blockDim.x) {
        const int2 b = bonds[h];
        for (int k = 0; k < nk; ++k) {
            const int l = wrap(lbase + k, Lt);
            const double c = ch[(size_t)l * Nh + h], s = sh[(size_t)l * Nh + h];
            double2 *row = U + (size_t)k * N;
            const double2 a = row[b.x], d = row[b.y];
            if (shi) {
                const double t = shi[(size_t)l * Nh + h];
                row[b.x] = make_double2(c * a.x + (s * d.x - t * d.y), c * a.y + (s * d.y + t * d.x));  // c a + s d
                row[b.y] = make_double2(c * d.x + (s * a.x + t * a.y), c * d.y + (s * a.y - t * a.x));  // c d + conj(s) a
            } else {
                row[b.x] = make_double2(c * a.x + s * d.x, c * a.y + s * d.y);
                row[b.y] = make_double2(c * d.x + s * a.x, c * d.y + s * a.y);
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void diag_stage(double2 *U, int nk, int N, int lbase, int Lt, const double *__restrict__ expV)
{
    for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        const double d = expV[(size_t)wrap(lbase + k, Lt) * N + i];
        double2 u = U[idx];
        U[idx] = make_double2(d * u.x, d * u.y);
    }
    __syncthreads();
}

// U[k] <- B_l U[k] (DAGGER = false) or B_lᴴ U[k] (true), l = (lbase + k) mod Lt.
// Sym:  B = Γ D Γᴴ (Hermitian);  Asym:  B = D Γ,  Bᴴ = Γᴴ D.   Γ = colours applied first-to-last.
template <bool SYM, bool DAGGER>
__device__ __forceinline__ void propagate(double2 *U, int nk, int lbase, const FdmArgs &a, const double *expV, const double *ch, const double *sh, const double *shi)
{
    if (SYM) {
        for (int c = a.ncol - 1; c >= 0; --c) colour_stage(U, nk, a.N, lbase, a.Lt, a.Nh, a.bonds, ch, sh, shi, a.col_off[c], a.col_off[c + 1]);
        diag_stage(U, nk, a.N, lbase, a.Lt, expV);
        for (int c = 0; c < a.ncol; ++c) colour_stage(U, nk, a.N, lbase, a.Lt, a.Nh, a.bonds, ch, sh, shi, a.col_off[c], a.col_off[c + 1]);
    } else if (!DAGGER) {
        for (int c = 0; c < a.ncol; ++c) colour_stage(U, nk, a.N, lbase, a.Lt, a.Nh, a.bonds, ch, sh, shi, a.col_off[c], a.col_off[c + 1]);
        diag_stage(U, nk, a.N, lbase, a.Lt, expV);
    } else {
        diag_stage(U, nk, a.N, lbase, a.Lt, expV);
        for (int c = a.ncol - 1; c >= 0; --c) colour_stage(U, nk, a.N, lbase, a.Lt, a.Nh, a.bonds, ch, sh, shi, a.col_off[c], a.col_off[c + 1]);
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// workgroup-wide complex sum; result valid in thread 0
__device__ __forceinline__ double2 block_sum(double2 v, double *red /* 2*8 doubles of LDS */)
{
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) { red[2 * wave] = v.x; red[2 * wave + 1] = v.y; }
    __syncthreads();
    double2 t = make_double2(0.0, 0.0);
    if (threadIdx.x == 0)
        for (int w = 0; w < nwave; ++w) { t.x += red[2 * w]; t.y += red[2 * w + 1]; }
    return t;
}

template <bool SYM, int OP>
__global__ void __launch_bounds__(kThreads) fdm_kernel(FdmArgs a)
{
    extern __shared__ double2 lds[];
    __shared__ double red[16];
    const int chunk = blockIdx.x % a.nchunk, sys = a.sys_first + blockIdx.x / a.nchunk;
    if (a.cg[sys].done) return;  // a.cg is never null (api_handle.hip, fdm_args / kpm_args: an all-zero state outside CG loops)
    const int w = sys / a.nrhs;
    const int Lt = a.Lt, N = a.N;
    const int l0 = chunk * a.Tc;
    const int nk = min(a.Tc, Lt - l0);
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double *ch = a.ch + (size_t)w * Lt * a.Nh, *sh = a.sh + (size_t)w * Lt * a.Nh;
    const double *shi = a.shi ? a.shi + (size_t)w * Lt * a.Nh : nullptr;
    const size_t sstride = (size_t)a.nsys * N;  // distance between time slices
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    double2 *U = a.scratch ? a.scratch + (size_t)blockIdx.x * a.scratch_stride : lds;
    double2 acc = make_double2(0.0, 0.0);

    if (OP == SMOQY_OP_M) {
        // (M v)[l] = v[l] -/+ B_l v[l-1]        (+ on l = first slice: antiperiodic wrap)
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N;
            U[idx] = in[(size_t)wrap(l0 + k - 1, Lt) * sstride + i];
        }
        __syncthreads();
        propagate<SYM, false>(U, nk, l0, a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = l0 + k;
            const double2 v = in[(size_t)l * sstride + i], u = U[idx];
            const double2 o = hopcomb_g(v, u, l == 0, false, a);
            if (a.partial) { acc.x += v.x * o.x + v.y * o.y; acc.y += v.x * o.y - v.y * o.x; }
            out[(size_t)l * sstride + i] = o;
        }
    } else if (OP == SMOQY_OP_MT) {
        // (Mᴴ v)[l] = v[l] -/+ B_{l+1}ᴴ v[l+1]   (+ on l = last slice)
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N;
            U[idx] = in[(size_t)wrap(l0 + k + 1, Lt) * sstride + i];
        }
        __syncthreads();
        propagate<SYM, true>(U, nk, wrap(l0 + 1, Lt), a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = l0 + k;
            const double2 v = in[(size_t)l * sstride + i], u = U[idx];
            const double2 o = hopcomb_g(v, u, l == Lt - 1, true, a);
            if (a.partial) { acc.x += v.x * o.x + v.y * o.y; acc.y += v.x * o.y - v.y * o.x; }
            out[(size_t)l * sstride + i] = o;
        }
    } else if (OP == SMOQY_OP_MTM) {
        // y = M v on slices l0 .. l0+nk (one halo slice), then out = Mᴴ y on l0 .. l0+nk-1
        double2 *Y = U + (size_t)(a.Tc + 1) * N;
        const int nk1 = nk + 1;
        for (int idx = threadIdx.x; idx < nk1 * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N;
            U[idx] = in[(size_t)wrap(l0 + k - 1, Lt) * sstride + i];
        }
        __syncthreads();
        propagate<SYM, false>(U, nk1, l0, a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk1 * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = wrap(l0 + k, Lt);
            const double2 v = in[(size_t)l * sstride + i], u = U[idx];
            Y[idx] = hopcomb_g(v, u, l == 0, false, a);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) U[idx] = Y[idx + N];
        __syncthreads();
        propagate<SYM, true>(U, nk, wrap(l0 + 1, Lt), a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = l0 + k;
            const double2 y = Y[idx], u = U[idx];
            const double2 o = hopcomb_g(y, u, l == Lt - 1, true, a);
            if (a.partial) {
                const double2 v = in[(size_t)l * sstride + i];
                acc.x += v.x * o.x + v.y * o.y;
                acc.y += v.x * o.y - v.y * o.x;
            }
            out[(size_t)l * sstride + i] = o;
        }
    } else {
        // MMᴴ: y = Mᴴ v on slices l0-1 .. l0+nk-1, then out = M y on l0 .. l0+nk-1
        double2 *Y = U + (size_t)(a.Tc + 1) * N;
        const int nk1 = nk + 1;
        for (int idx = threadIdx.x; idx < nk1 * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N;
            U[idx] = in[(size_t)wrap(l0 + k, Lt) * sstride + i];
        }
        __syncthreads();
        propagate<SYM, true>(U, nk1, l0, a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk1 * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = wrap(l0 + k - 1, Lt);
            const double2 v = in[(size_t)l * sstride + i], u = U[idx];
            Y[idx] = hopcomb_g(v, u, l == Lt - 1, true, a);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) U[idx] = Y[idx];
        __syncthreads();
        propagate<SYM, false>(U, nk, l0, a, expV, ch, sh, shi);
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N, l = l0 + k;
            const double2 y = Y[idx + N], u = U[idx];
            const double2 o = hopcomb_g(y, u, l == 0, false, a);
            if (a.partial) {
                const double2 v = in[(size_t)l * sstride + i];
                acc.x += v.x * o.x + v.y * o.y;
                acc.y += v.x * o.y - v.y * o.x;
            }
            out[(size_t)l * sstride + i] = o;
        }
    }
    if (a.partial) {
        const double2 t = block_sum(acc, red);
        if (threadIdx.x == 0) a.partial[(size_t)sys * a.nchunk + chunk] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// stand-alone checkerboard_lmul! / checkerboard_ldiv! on a colour interval
// (src/checkerboard_matrix_multiply.jl:26-72, 98-145): in place on a device vector, one workgroup per
// (tau-chunk, system), slices staged through LDS exactly like the fused kernels.
// lmul, not transposed: colours first..last;  transposed: last..first (:45-47)
// ldiv (factor [[c,-s],[-s,c]]), not transposed: last..first (:118-120);  transposed: first..last
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) checkerboard_kernel(FdmArgs a, int inverse, int transposed, int col0, int ncols)
{
    extern __shared__ double2 lds_cb[];
    double2 *U = a.scratch ? a.scratch + (size_t)blockIdx.x * a.scratch_stride : lds_cb;
    const int chunk = blockIdx.x % a.nchunk, sys = a.sys_first + blockIdx.x / a.nchunk;
    const int w = sys / a.nrhs, Lt = a.Lt, N = a.N;
    const int l0 = chunk * a.Tc, nk = min(a.Tc, Lt - l0);
    const double *ch = a.ch + (size_t)w * Lt * a.Nh, *sh = a.sh + (size_t)w * Lt * a.Nh;
    const double *shi = a.shi ? a.shi + (size_t)w * Lt * a.Nh : nullptr;
    const size_t sstride = (size_t)a.nsys * N;
    double2 *v = a.out + (size_t)sys * N;
    for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        U[idx] = v[(size_t)(l0 + k) * sstride + i];
    }
    __syncthreads();
    const bool reversed = (transposed != 0) != (inverse != 0);
    for (int q = 0; q < ncols; ++q) {
        const int c = reversed ? col0 + ncols - 1 - q : col0 + q;
        const int cb = a.col_off[c], ce = a.col_off[c + 1];
        for (int h = cb + (int)threadIdx.x; h < ce; h += (int)blockDim.x) {
            const int2 b = a.bonds[h];
            for (int k = 0; k < nk; ++k) {
                const double cc = ch[(size_t)(l0 + k) * a.Nh + h];
                const double ss = inverse ? -sh[(size_t)(l0 + k) * a.Nh + h] : sh[(size_t)(l0 + k) * a.Nh + h];
                double2 *row = U + (size_t)k * N;
                const double2 x = row[b.x], y = row[b.y];
                if (shi) {  // complex hopping: [[c, s], [conj(s), c]], inverse [[c, -s], [-conj(s), c]] (:133-141)
                    const double tt = inverse ? -shi[(size_t)(l0 + k) * a.Nh + h] : shi[(size_t)(l0 + k) * a.Nh + h];
                    row[b.x] = make_double2(cc * x.x + (ss * y.x - tt * y.y), cc * x.y + (ss * y.y + tt * y.x));
                    row[b.y] = make_double2(cc * y.x + (ss * x.x + tt * x.y), cc * y.y + (ss * x.y - tt * x.x));
                    continue;
                }
                row[b.x] = make_double2(cc * x.x + ss * y.x, cc * x.y + ss * y.y);
                row[b.y] = make_double2(cc * y.x + ss * x.x, cc * y.y + ss * x.y);
            }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        v[(size_t)(l0 + k) * sstride + i] = U[idx];
    }
}

void launch_checkerboard(hipStream_t st, const FdmArgs &a, int inverse, int transposed, int col0, int ncols)
{
    const size_t lds = a.scratch ? 0 : sizeof(double2) * (size_t)a.N * (size_t)a.Tc;
    hipLaunchKernelGGL(checkerboard_kernel, dim3((unsigned)(a.nchunk * a.sys_count)), dim3(kThreads), lds, st, a, inverse, transposed, col0, ncols);
}

size_t fdm_lds_bytes(int op, int N, int Tc)
{
    const bool fused = (op == SMOQY_OP_MTM || op == SMOQY_OP_MMT);
    return sizeof(double2) * (size_t)N * (size_t)(fused ? 2 * (Tc + 1) : Tc);
}

template <bool SYM, int OP>
static void launch_one(hipStream_t st, const FdmArgs &a, size_t lds)
{
    hipLaunchKernelGGL((fdm_kernel<SYM, OP>), dim3((unsigned)(a.nchunk * a.sys_count)), dim3(kThreads), lds, st, a);
}

// raise the dynamic-LDS limit of every instantiation once, outside any stream capture
template <bool SYM, int OP>
static void configure_one(hipError_t &first, const char **what) { SMOQY_SET_LDS((fdm_kernel<SYM, OP>), 160 * 1024 - 256); }
hipError_t configure_fdm_kernels(const char **what)
{
    hipError_t first = hipSuccess;
    SMOQY_SET_LDS(checkerboard_kernel, 160 * 1024 - 256);
    configure_one<true, 0>(first, what); configure_one<true, 1>(first, what); configure_one<true, 2>(first, what); configure_one<true, 3>(first, what);
    configure_one<false, 0>(first, what); configure_one<false, 1>(first, what); configure_one<false, 2>(first, what); configure_one<false, 3>(first, what);
    return first;
}

void launch_fdm(hipStream_t st, int op, bool sym, const FdmArgs &a, size_t lds)
{
    if (sym) {
        switch (op) {
            case SMOQY_OP_M: launch_one<true, SMOQY_OP_M>(st, a, lds); break;
            case SMOQY_OP_MT: launch_one<true, SMOQY_OP_MT>(st, a, lds); break;
            case SMOQY_OP_MTM: launch_one<true, SMOQY_OP_MTM>(st, a, lds); break;
            default: launch_one<true, SMOQY_OP_MMT>(st, a, lds); break;
        }
    } else {
        switch (op) {
            case SMOQY_OP_M: launch_one<false, SMOQY_OP_M>(st, a, lds); break;
            case SMOQY_OP_MT: launch_one<false, SMOQY_OP_MT>(st, a, lds); break;
            case SMOQY_OP_MTM: launch_one<false, SMOQY_OP_MTM>(st, a, lds); break;
            default: launch_one<false, SMOQY_OP_MMT>(st, a, lds); break;
        }
    }
}

}  // namespace smoqy
