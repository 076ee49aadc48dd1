// Layout conversion at the C-ABI boundary, field refresh, Λ applies, dot products, FFT
// twiddles and the BLAS-1 side of the on-device conjugate gradient (gfx950).
//
// Reference semantics cited per kernel.  All reductions are deterministic: fixed-order
// per-workgroup partials (wavefront __shfl_down trees + one LDS hop) that every consumer
// workgroup re-reduces in the same order, so no atomics and no run-to-run jitter.
#include "smoqy_internal.h"

#include <algorithm>

namespace smoqy {

__device__ __forceinline__ double wsum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// workgroup sum of a (re, im) pair, broadcast to every thread
__device__ __forceinline__ double2 block_sum_bcast(double2 v, double *red /* >= 18 doubles */)
{
    v.x = wsum(v.x);
    v.y = wsum(v.y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
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

// every thread gets sum_{c<n} part[c] (complex) in a fixed order
__device__ __forceinline__ double2 reduce_partials(const double2 *part, int n, double *red)
{
    double2 t = make_double2(0.0, 0.0);
    for (int c = threadIdx.x; c < n; c += blockDim.x) { t.x += part[c].x; t.y += part[c].y; }
    return block_sum_bcast(t, red);
}

__device__ __forceinline__ double reduce_partials(const double *part, int n, double *red)
{
    double2 t = make_double2(0.0, 0.0);
    for (int c = threadIdx.x; c < n; c += blockDim.x) t.x += part[c];
    return block_sum_bcast(t, red).x;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cdiv(double2 a, double2 b)
{
    const double d = b.x * b.x + b.y * b.y;
    return make_double2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// ---------------------------------------------------------------------------------------------
// layout conversion: reference (tau fastest, Ltau x N per system) <-> device [l][s][i]
// 32 x 32 tiles through LDS so both sides are coalesced
// ---------------------------------------------------------------------------------------------
template <typename T, bool TO_DEVICE>
__global__ void transpose_kernel(const T *__restrict__ src, T *__restrict__ dst, int Lt, int N, int dev_slice_stride /* nsys*N or n */, int dev_off /* sys*N */, size_t host_sys_stride /* Lt*N */)
{
    __shared__ T tile[32][33];
    const int sysl = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int l0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    const T *hsrc = TO_DEVICE ? src + (size_t)sysl * host_sys_stride : nullptr;
    T *hdst = TO_DEVICE ? nullptr : dst + (size_t)sysl * host_sys_stride;
    const size_t doff = (size_t)dev_off + (size_t)sysl * N;
    if (TO_DEVICE) {
        for (int r = ty; r < 32; r += 8) {  // r: site within tile, tx: slice within tile
            const int i = i0 + r, l = l0 + tx;
            if (i < N && l < Lt) tile[r][tx] = hsrc[(size_t)i * Lt + l];
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {  // r: slice within tile, tx: site within tile
            const int l = l0 + r, i = i0 + tx;
            if (i < N && l < Lt) dst[(size_t)l * dev_slice_stride + doff + i] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            const int l = l0 + r, i = i0 + tx;
            if (i < N && l < Lt) tile[tx][r] = src[(size_t)l * dev_slice_stride + doff + i];
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, l = l0 + tx;
            if (i < N && l < Lt) hdst[(size_t)i * Lt + l] = tile[r][tx];
        }
    }
}

void launch_transpose_in(hipStream_t st, const double2 *h, double2 *d, int Lt, int N, int nsys, int sys0, int count)
{
    dim3 g((Lt + 31) / 32, (N + 31) / 32, count);
    hipLaunchKernelGGL((transpose_kernel<double2, true>), g, dim3(256), 0, st, h, d, Lt, N, nsys * N, sys0 * N, (size_t)Lt * N);
}
void launch_transpose_out(hipStream_t st, const double2 *d, double2 *h, int Lt, int N, int nsys, int sys0, int count)
{
    dim3 g((Lt + 31) / 32, (N + 31) / 32, count);
    hipLaunchKernelGGL((transpose_kernel<double2, false>), g, dim3(256), 0, st, d, h, Lt, N, nsys * N, sys0 * N, (size_t)Lt * N);
}
void launch_transpose_real_in(hipStream_t st, const double *src, double *dst, int Lt, int n)
{
    dim3 g((Lt + 31) / 32, (n + 31) / 32, 1);
    hipLaunchKernelGGL((transpose_kernel<double, true>), g, dim3(256), 0, st, src, dst, Lt, n, n, 0, (size_t)0);
}
void launch_transpose_real_out(hipStream_t st, const double *src, double *dst, int Lt, int n)
{
    dim3 g((Lt + 31) / 32, (n + 31) / 32, 1);
    hipLaunchKernelGGL((transpose_kernel<double, false>), g, dim3(256), 0, st, src, dst, Lt, n, n, 0, (size_t)0);
}

// ---------------------------------------------------------------------------------------------
// update!(fdm, fpi) — src/FermionDetMatrix.jl:208-236.  V is N x Ltau, t is Nh x Ltau (both
// site/bond fastest == already slice-major); outputs slice-major.
// ---------------------------------------------------------------------------------------------
__global__ void fields_kernel(const double *__restrict__ V, const double *__restrict__ t, const int *__restrict__ perm0, double *__restrict__ expV, double *__restrict__ ch,
                              double *__restrict__ sh, int Lt, int N, int Nh, double dtau, double dtau_k)
{
    // Lt may be nwalkers * Ltau: the walkers' arrays are contiguous on both sides.  V == nullptr or
    // t == nullptr skips that part (fields unchanged).
    const size_t nV = V ? (size_t)Lt * N : 0, nT = t ? (size_t)Lt * Nh : 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nV + nT; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nV) {
            expV[idx] = exp(-dtau * V[idx]);  // :217
        } else {
            const size_t j = idx - nV;
            const int l = (int)(j / Nh), h = (int)(j - (size_t)l * Nh);
            const double tt = t[(size_t)l * Nh + perm0[h]];  // :224-228
            const double a = dtau_k * fabs(tt);
            ch[j] = cosh(a);                                                    // :230
            sh[j] = (tt > 0 ? 1.0 : (tt < 0 ? -1.0 : 0.0)) * sinh(a);           // :231
        }
    }
}

// T = ComplexF64: the hopping is complex128, sinh = sign(conj t) sinh(Δτ′|t|) = conj(t)/|t| · sinh (Julia's sign(0) = 0)
__global__ void fields_kernel_c(const double *__restrict__ V, const double2 *__restrict__ t, const int *__restrict__ perm0, double *__restrict__ expV, double *__restrict__ ch,
                                double *__restrict__ sh, double *__restrict__ shi, int Lt, int N, int Nh, double dtau, double dtau_k)
{
    const size_t nV = V ? (size_t)Lt * N : 0, nT = t ? (size_t)Lt * Nh : 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nV + nT; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nV) {
            expV[idx] = exp(-dtau * V[idx]);  // :217
        } else {
            const size_t j = idx - nV;
            const int l = (int)(j / Nh), h = (int)(j - (size_t)l * Nh);
            const double2 tt = t[(size_t)l * Nh + perm0[h]];  // :224-228
            const double ab = hypot(tt.x, tt.y), a = dtau_k * ab, sn = sinh(a);
            ch[j] = cosh(a);                                   // :230
            sh[j] = ab > 0.0 ? tt.x / ab * sn : 0.0;           // :231, Re sign(conj t)
            shi[j] = ab > 0.0 ? -tt.y / ab * sn : 0.0;         //        Im sign(conj t)
        }
    }
}

void launch_fields_from_path_integral_c(hipStream_t st, const double *V, const double2 *t, const int *perm0, double *expV, double *ch, double *sh, double *shi, int Lt, int N, int Nh, double dtau, double dtau_k)
{
    const size_t tot = (V ? (size_t)Lt * N : 0) + (t ? (size_t)Lt * Nh : 0);
    if (tot == 0) return;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fields_kernel_c, dim3(blocks), dim3(256), 0, st, V, t, perm0, expV, ch, sh, shi, Lt, N, Nh, dtau, dtau_k);
}

void launch_fields_from_path_integral(hipStream_t st, const double *V, const double *t, const int *perm0, double *expV, double *ch, double *sh, int Lt, int N, int Nh, double dtau, double dtau_k)
{
    const size_t tot = (V ? (size_t)Lt * N : 0) + (t ? (size_t)Lt * Nh : 0);
    if (tot == 0) return;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fields_kernel, dim3(blocks), dim3(256), 0, st, V, t, perm0, expV, ch, sh, Lt, N, Nh, dtau, dtau_k);
}

// ---------------------------------------------------------------------------------------------
// update_Λ! — src/holstein_shift_matrix.jl:2-44  (x is Nph x Ltau, phonon fastest)
// ---------------------------------------------------------------------------------------------
__global__ void lambda_init_kernel(double *Lam, int Lt, int N, int Lt1)
{
    // Lt = nwalkers * Lt1 slices in a row; the first slice of every walker is +1 (:11-12)
    const size_t tot = (size_t)Lt * N, per = (size_t)Lt1 * N;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) Lam[idx] = (idx % per < (size_t)N) ? 1.0 : -1.0;
}

__global__ void lambda_couple_kernel(double *Lam, int Lt, int N, const double *__restrict__ x, int Nph, double dtau, int ncoup, const int *c2p, const int *c2s, const double *alpha, const double *alpha3,
                                     const int *phsym, const int *site_first, const int *site_next)
{
    // one thread per (slice, site); the couplings of a site are walked through a linked list
    // (site_first / site_next) in coupling order, exactly the order of the reference loop (:17-40)
    const size_t tot = (size_t)Lt * N;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / N), i = (int)(idx - (size_t)l * N);
        double v = Lam[idx];
        for (int c = site_first[i]; c >= 0; c = site_next[c]) {
            if (!phsym[c]) continue;
            const double xp = x[(size_t)l * Nph + c2p[c]];
            v *= exp(dtau * (alpha[c] * xp + alpha3[c] * xp * xp * xp) / 2);  // :37
        }
        Lam[idx] = v;
    }
}

void launch_lambda_update(hipStream_t st, double *Lam, int Lt, int N, const double *x, int Nph, double dtau, int ncoup, const int *c2p, const int *c2s, const double *alpha, const double *alpha3, const int *phsym,
                          const int *site_first, const int *site_next, int Lt1)
{
    const size_t tot = (size_t)Lt * N;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lambda_init_kernel, dim3(blocks), dim3(256), 0, st, Lam, Lt, N, Lt1);
    if (ncoup > 0) hipLaunchKernelGGL(lambda_couple_kernel, dim3(blocks), dim3(256), 0, st, Lam, Lt, N, x, Nph, dtau, ncoup, c2p, c2s, alpha, alpha3, phsym, site_first, site_next);
}

// ---------------------------------------------------------------------------------------------
// Λ applies — src/holstein_shift_matrix.jl:47-71, 74-98, 102-126, 129-153.
// In slice-major layout all four are one-slice shifts with a diagonal scale; out-of-place here,
// the host wrapper handles out == in by ping-ponging buffers.
// ---------------------------------------------------------------------------------------------
__global__ void lambda_apply_kernel(int op, double2 *__restrict__ out, const double2 *__restrict__ in, const double *__restrict__ Lam, int Lt, int N, int nsys, int nrhs, int wfixed)
{
    const size_t per_slice = (size_t)nsys * N, tot = per_slice * Lt;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per_slice);
        const size_t rem = idx - (size_t)l * per_slice;
        const int s = (int)(rem / N), i = (int)(rem - (size_t)s * N);
        const int w = wfixed >= 0 ? wfixed : s / nrhs;
        const double *L = Lam + (size_t)w * Lt * N;
        double2 o;
        if (op == SMOQY_LAMBDA_MUL) {  // (Λv)[l] = Λ[l+1] v[l+1], wrap to slice 0
            const int lp = (l + 1 == Lt) ? 0 : l + 1;
            const double f = L[(size_t)lp * N + i];
            const double2 v = in[(size_t)lp * per_slice + rem];
            o = make_double2(f * v.x, f * v.y);
        } else if (op == SMOQY_LAMBDA_LDIV) {  // (Λ⁻¹v)[l] = v[l-1] / Λ[l]
            const int lm = (l == 0) ? Lt - 1 : l - 1;
            const double f = L[(size_t)l * N + i];
            const double2 v = in[(size_t)lm * per_slice + rem];
            o = make_double2(v.x / f, v.y / f);
        } else if (op == SMOQY_LAMBDA_MULT) {  // (Λᵀv)[l] = Λ[l] v[l-1]
            const int lm = (l == 0) ? Lt - 1 : l - 1;
            const double f = L[(size_t)l * N + i];
            const double2 v = in[(size_t)lm * per_slice + rem];
            o = make_double2(f * v.x, f * v.y);
        } else {  // (Λ⁻ᵀv)[l] = v[l+1] / Λ[l+1]
            const int lp = (l + 1 == Lt) ? 0 : l + 1;
            const double f = L[(size_t)lp * N + i];
            const double2 v = in[(size_t)lp * per_slice + rem];
            o = make_double2(v.x / f, v.y / f);
        }
        out[idx] = o;
    }
}

void launch_lambda_apply(hipStream_t st, int op, double2 *out, const double2 *in, const double *Lam, int Lt, int N, int nsys, int nrhs, int wfixed)
{
    const size_t tot = (size_t)Lt * nsys * N;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(lambda_apply_kernel, dim3(blocks), dim3(256), 0, st, op, out, in, Lam, Lt, N, nsys, nrhs, wfixed);
}

// real <-> complex staging of the real-vector entry points (KPMPreconditioner.jl:306 `mul!(v, U, u)` with real u, :344 `real(v)`)
__global__ void real_to_complex_kernel(const double *__restrict__ re, double2 *__restrict__ z, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) z[i] = make_double2(re[i], 0.0);
}
__global__ void complex_to_real_kernel(const double2 *__restrict__ z, double *__restrict__ re, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) re[i] = z[i].x;
}
void launch_real_to_complex(hipStream_t st, const double *re, double2 *z, size_t n)
{
    hipLaunchKernelGGL(real_to_complex_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, re, z, n);
}
void launch_complex_to_real(hipStream_t st, const double2 *z, double *re, size_t n)
{
    hipLaunchKernelGGL(complex_to_real_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, st, z, re, n);
}

// ---------------------------------------------------------------------------------------------
// device stream copy: the measured HBM ceiling bench.py quotes next to the 8 TB/s specification.
// Shape chosen by tools/copy_probe.hip on MI355X (1 GiB -> 1 GiB): a workgroup owns contiguous 16 KiB tiles, four independent
// 16-byte nontemporal loads in flight per lane, 16384 workgroups: 6.31 TB/s; a grid-stride loop whose four loads sit 16 MiB apart
// (same channel) reaches 4.3-4.9 TB/s, hipMemcpyDtoD 5.05 TB/s.
// ---------------------------------------------------------------------------------------------
typedef double v2d_t __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) stream_copy_kernel(v2d_t *__restrict__ dst, const v2d_t *__restrict__ src, size_t n)
{
    constexpr int U = 4;
    const size_t tile = (size_t)U * 256, ntile = n / tile;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const size_t base = t * tile + threadIdx.x;
        v2d_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&src[base + u * 256]);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &dst[base + u * 256]);
    }
    for (size_t i = ntile * tile + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

void launch_stream_copy(hipStream_t st, double2 *dst, const double2 *src, size_t n)
{
    hipLaunchKernelGGL(stream_copy_kernel, dim3(16384), dim3(256), 0, st, reinterpret_cast<v2d_t *>(dst), reinterpret_cast<const v2d_t *>(src), n);
}

// ---------------------------------------------------------------------------------------------
// dot(a, b) per system (LinearAlgebra.dot: conjugate-linear in a)
// ---------------------------------------------------------------------------------------------
__global__ void dot_partial_kernel(const double2 *__restrict__ a, const double2 *__restrict__ b, double2 *__restrict__ partial, int Lt, int N, int nsys, int Tc, int nchunk)
{
    __shared__ double red[18];
    const int chunk = blockIdx.x % nchunk, sys = blockIdx.x / nchunk;
    const int l0 = chunk * Tc, nk = min(Tc, Lt - l0);
    const size_t sstride = (size_t)nsys * N;
    double2 acc = make_double2(0.0, 0.0);
    for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N;
        const size_t off = (size_t)(l0 + k) * sstride + (size_t)sys * N + i;
        const double2 x = a[off], y = b[off];
        acc.x += x.x * y.x + x.y * y.y;
        acc.y += x.x * y.y - x.y * y.x;
    }
    const double2 t = block_sum_bcast(acc, red);
    if (threadIdx.x == 0) partial[(size_t)sys * nchunk + chunk] = t;
}

__global__ void dot_final_kernel(const double2 *__restrict__ partial, double2 *__restrict__ out, int nchunk)
{
    __shared__ double red[18];
    const double2 t = reduce_partials(partial + (size_t)blockIdx.x * nchunk, nchunk, red);
    if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// second half of launch_dot alone: the per-chunk partials were written by another kernel (cg_finish with the fused Λ⁻¹ and Φ·Ψ)
void launch_dot_final(hipStream_t st, const double2 *partial, double2 *out, int nsys, int nchunk)
{
    hipLaunchKernelGGL(dot_final_kernel, dim3(nsys), dim3(64), 0, st, partial, out, nchunk);
}

void launch_dot(hipStream_t st, const double2 *a, const double2 *b, double2 *partial, double2 *out, int Lt, int N, int nsys, int Tc, int nchunk)
{
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nchunk * nsys), dim3(kThreads), 0, st, a, b, partial, Lt, N, nsys, Tc, nchunk);
    hipLaunchKernelGGL(dot_final_kernel, dim3(nsys), dim3(64), 0, st, partial, out, nchunk);
}

// ---------------------------------------------------------------------------------------------
// FourierTransformer twiddle — src/FourierTransformer.jl:15, 46, 61:  θ_l = exp(-iπ l/Lτ).
// forward:  v[l] *= θ_l/√Lτ  (before the FFT);  inverse:  v[l] *= conj(θ_l)/√Lτ  (after the
// unnormalised inverse FFT, i.e. (1/Lτ)·√Lτ/θ_l).
// ---------------------------------------------------------------------------------------------
__global__ void make_twiddle_kernel(double2 *tw, int Lt, double f)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= Lt) return;
    double s, c;
    sincospi(-(double)l / (double)Lt, &s, &c);
    tw[l] = make_double2(c * f, s * f);
}

// tw[l] = scale * exp(-iπ l / Lτ)
void launch_make_twiddle(hipStream_t st, double2 *tw, int Lt, double scale) { hipLaunchKernelGGL(make_twiddle_kernel, dim3((Lt + 63) / 64), dim3(64), 0, st, tw, Lt, scale); }

__device__ __forceinline__ double2 twiddle(const double2 *__restrict__ tw, int l, bool conj_)
{
    const double2 t = tw[l];
    return conj_ ? make_double2(t.x, -t.y) : t;
}

__global__ void twiddle_kernel(double2 *v, const double2 *__restrict__ tw, int Lt, int N, int nsys, int inverse)
{
    const size_t per_slice = (size_t)nsys * N, tot = per_slice * Lt;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per_slice);
        v[idx] = cmul(v[idx], twiddle(tw, l, inverse != 0));
    }
}

void launch_fft_twiddle(hipStream_t st, double2 *v, const double2 *tw, int Lt, int N, int nsys, int inverse)
{
    const size_t tot = (size_t)Lt * nsys * N;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(twiddle_kernel, dim3(blocks), dim3(256), 0, st, v, tw, Lt, N, nsys, inverse);
}

// ---------------------------------------------------------------------------------------------
// conjugate gradient, BLAS-1 side — src/IterativeSolvers/ConjugateGradient.jl:93-249.
//
// The loop runs in the "twiddled" basis ṽ[l] = θ_l v[l], θ_l = exp(-iπ l/Lτ) (the FourierTransformer
// phase, src/FourierTransformer.jl:15).  Θ is unitary and diagonal in τ, so every inner product and
// the whole CG recurrence are unchanged, while
//   * the operator Θ MᵀM Θᴴ is the same checkerboard kernel with a uniform hop phase exp(-iπ/Lτ) and
//     a PERIODIC time direction (FdmArgs::hop, antiperiodic = 0);
//   * the preconditioner U⁻¹ P̂ U loses both twiddle passes: P̃⁻¹ = FFT⁻¹ · P̂ · FFT, rocFFT reads r̃
//     directly (out of place) and its output IS z̃;
//   * r·z is formed by Parseval inside the Chebyshev kernel (it holds r̂_ω and ẑ_ω), so no pass over
//     r and z is needed for it.
// b is twiddled on the way in (cg_init), x on the way out (cg_finish).
// One workgroup per (tau-chunk, system); per-system scalars never leave the device.
// ---------------------------------------------------------------------------------------------
#define CG_PROLOGUE                                                        \
    __shared__ double red[18];                                             \
    const int chunk = blockIdx.x % a.nchunk, sys = blockIdx.x / a.nchunk; \
    const int l0 = chunk * a.Tc, nk = min(a.Tc, a.Lt - l0);               \
    const size_t sstride = (size_t)a.nsys * a.N;                           \
    const size_t base = (size_t)sys * a.N;                                 \
    const size_t pidx = (size_t)sys * a.nchunk + chunk;                    \
    (void)red; (void)l0; (void)nk; (void)sstride; (void)base; (void)pidx;

// :108-121 — r̃0 = Θb (x = 0) or Θb - z with z = Ã x̃0 from the MᵀM kernel; partial |b|², |r|²
template <bool X_IS_B>
__global__ void __launch_bounds__(kThreads) cg_init_kernel(CgArgs a)
{
    CG_PROLOGUE
    double2 acc = make_double2(0.0, 0.0);
    for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
        const int k = idx / a.N, i = idx - k * a.N, l = l0 + k;
        const size_t off = (size_t)l * sstride + base + i;
        double2 bsrc;
        if (a.lam) {  // b = Λ⁻ᵀΦ: (Λ⁻ᵀv)[l] = v[l+1] / Λ[l+1] (src/holstein_shift_matrix.jl:129-153; the arithmetic of lambda_apply_kernel)
            const int lp = (l + 1 == a.Lt) ? 0 : l + 1;
            const double f = a.lam[((size_t)(sys / a.nrhs) * a.Lt + lp) * a.N + i];
            const double2 v = a.phi[(size_t)lp * sstride + base + i];
            bsrc = make_double2(v.x / f, v.y / f);
        } else {
            bsrc = a.b[off];
        }
        const double2 bv = cmul(bsrc, a.th[l]);
        double2 rv;
        if (X_IS_B) {
            rv = bv;
            a.x[off] = make_double2(0.0, 0.0);
        } else {
            const double2 zv = a.z[off];
            rv = make_double2(bv.x - zv.x, bv.y - zv.y);
        }
        a.r[off] = rv;
        acc.x += bv.x * bv.x + bv.y * bv.y;
        acc.y += rv.x * rv.x + rv.y * rv.y;
    }
    const double2 t = block_sum_bcast(acc, red);
    if (threadIdx.x == 0) { a.part_bb[pidx] = t.x; a.part_rr[pidx] = t.y; }
}

// :123-134 / :199-212 — p0 = z0 (= P⁻¹ r0, already in v, or r0), eps0, early exit
__global__ void __launch_bounds__(kThreads) cg_start_kernel(CgArgs a)
{
    CG_PROLOGUE
    const double bb = reduce_partials(a.part_bb + (size_t)sys * a.nchunk, a.nchunk, red);
    const double rr = reduce_partials(a.part_rr + (size_t)sys * a.nchunk, a.nchunk, red);
    // non-finite input: report NaN and stop (the host turns it into an error); b = 0: x = 0 is exact
    const bool bad = !(bb == bb) || !(rr == rr) || isinf(bb) || isinf(rr);
    const double eps = bad ? nan("") : ((bb > 0.0) ? sqrt(rr) / sqrt(bb) : 0.0);
    const bool conv = bad || !(bb > 0.0) || eps < a.tol;
    if (!conv) {
        const double2 *zz = a.use_precond ? a.v : a.r;
        for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
            const int k = idx / a.N, i = idx - k * a.N;
            const size_t off = (size_t)(l0 + k) * sstride + base + i;
            a.p[off] = zz[off];
        }
    }
    if (threadIdx.x == 0) {
        if (!a.use_precond) a.part_rz[(size_t)sys * a.rz_stride + chunk] = make_double2(a.part_rr[pidx], 0.0);  // r·z = r·r
        if (chunk == 0) {
            CgState &s = a.st[sys];
            s.normb2 = bb;
            s.eps = eps;
            s.iters = 0;
            s.done = bad ? 2 : (conv ? 1 : 0);
            s.stop = s.done;
        }
    }
}

// :219-226 — α = (r·z)/(p·Ap);  x += α p;  r -= α Ap;  partial |r|²
__global__ void __launch_bounds__(kThreads) cg_update_xr_kernel(CgArgs a)
{
    CG_PROLOGUE
    {   // latch `done` (written by an earlier launch) into `stop` for this iteration's cg_update_p, see CgState::stop
        const int done = a.st[sys].done;
        if (done) {
            if (chunk == 0 && threadIdx.x == 0) a.st[sys].stop = done;
            return;
        }
    }
    const double2 rz = reduce_partials(a.part_rz + (size_t)sys * a.rz_stride, a.nrz, red);
    const double2 pz = reduce_partials(a.part_pz + (size_t)sys * a.nchunk, a.nchunk, red);
    const double2 alpha = cdiv(rz, pz);
    double acc = 0.0;
    for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
        const int k = idx / a.N, i = idx - k * a.N;
        const size_t off = (size_t)(l0 + k) * sstride + base + i;
        const double2 pv = a.p[off], zv = a.z[off];
        double2 xv = a.x[off], rv = a.r[off];
        const double2 ap = cmul(alpha, pv), az = cmul(alpha, zv);
        xv.x += ap.x; xv.y += ap.y;
        rv.x -= az.x; rv.y -= az.y;
        a.x[off] = xv;
        a.r[off] = rv;
        acc += rv.x * rv.x + rv.y * rv.y;
    }
    const double2 t = block_sum_bcast(make_double2(acc, 0.0), red);
    if (threadIdx.x == 0) {
        a.part_rr[pidx] = t.x;
        if (chunk == 0) { a.st[sys].rho_re = rz.x; a.st[sys].rho_im = rz.y; }
    }
}

// :229-245 — eps = |r|/|b| and the stop test (on the unpreconditioned residual, before the β
// update), then β = (r·z)_new/(r·z)_old and p = z + β p with z = P⁻¹ r (in v) or r
__global__ void __launch_bounds__(kThreads) cg_update_p_kernel(CgArgs a)
{
    CG_PROLOGUE
    if (a.st[sys].stop) return;  // not `done`: chunk 0 of this very launch writes it
    const double rr = reduce_partials(a.part_rr + (size_t)sys * a.nchunk, a.nchunk, red);
    const double eps = sqrt(rr) / sqrt(a.st[sys].normb2);
    const bool conv = eps < a.st[sys].tol;
    if (!conv) {
        const double2 rz = a.use_precond ? reduce_partials(a.part_rz + (size_t)sys * a.rz_stride, a.nrz, red) : make_double2(rr, 0.0);
        const double2 beta = cdiv(rz, make_double2(a.st[sys].rho_re, a.st[sys].rho_im));
        const double2 *zz = a.use_precond ? a.v : a.r;
        for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
            const int k = idx / a.N, i = idx - k * a.N;
            const size_t off = (size_t)(l0 + k) * sstride + base + i;
            const double2 zv = zz[off], bp = cmul(beta, a.p[off]);
            a.p[off] = make_double2(zv.x + bp.x, zv.y + bp.y);
        }
    }
    if (threadIdx.x == 0) {
        if (!a.use_precond && !conv) a.part_rz[(size_t)sys * a.rz_stride + chunk] = make_double2(a.part_rr[pidx], 0.0);
        if (chunk == 0) {
            CgState &s = a.st[sys];
            s.eps = eps;
            s.iters += 1;
            if (conv) s.done = 1;
            else if (s.iters >= s.maxiter) s.done = 2;
        }
    }
}

// back to the reference basis: x = Θᴴ x̃
__global__ void __launch_bounds__(kThreads) cg_finish_kernel(CgArgs a)
{
    CG_PROLOGUE
    if (a.lam) {
        // Ψ[l] = (Λ⁻¹ x)[l] = x[l-1] / Λ[l] with x = Θᴴ x̃ (src/holstein_shift_matrix.jl:74-98, PFFCalculator.jl:107), out of place, and the
        // partial of Φ·Ψ (:109) over this chunk: the operations and the summation order of cg_finish + lambda_apply + dot_partial
        double2 acc = make_double2(0.0, 0.0);
        for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
            const int k = idx / a.N, i = idx - k * a.N, l = l0 + k;
            const int lm = (l == 0) ? a.Lt - 1 : l - 1;
            const size_t off = (size_t)l * sstride + base + i;
            const double2 t = a.th[lm];
            const double2 xv = cmul(a.x[(size_t)lm * sstride + base + i], make_double2(t.x, -t.y));
            const double f = a.lam[((size_t)(sys / a.nrhs) * a.Lt + l) * a.N + i];
            const double2 y = make_double2(xv.x / f, xv.y / f);
            a.x_out[off] = y;
            const double2 p = a.phi[off];
            acc.x += p.x * y.x + p.y * y.y;
            acc.y += p.x * y.y - p.y * y.x;
        }
        const double2 tsum = block_sum_bcast(acc, red);
        if (threadIdx.x == 0) a.part_dot[pidx] = tsum;
        return;
    }
    for (int idx = threadIdx.x; idx < nk * a.N; idx += blockDim.x) {
        const int k = idx / a.N, i = idx - k * a.N, l = l0 + k;
        const size_t off = (size_t)l * sstride + base + i;
        const double2 t = a.th[l];
        a.x[off] = cmul(a.x[off], make_double2(t.x, -t.y));
    }
}

void launch_cg_init(hipStream_t s, const CgArgs &a, bool x_is_b)
{
    if (x_is_b) hipLaunchKernelGGL((cg_init_kernel<true>), dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a);
    else hipLaunchKernelGGL((cg_init_kernel<false>), dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a);
}
void launch_cg_start(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(cg_start_kernel, dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a); }
void launch_cg_update_xr(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(cg_update_xr_kernel, dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a); }
void launch_cg_update_p(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(cg_update_p_kernel, dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a); }
void launch_cg_finish(hipStream_t s, const CgArgs &a) { hipLaunchKernelGGL(cg_finish_kernel, dim3(a.nchunk * a.nsys), dim3(kThreads), 0, s, a); }

}  // namespace smoqy
