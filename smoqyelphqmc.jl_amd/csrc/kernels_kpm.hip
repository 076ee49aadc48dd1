// KPM preconditioner kernels for gfx950: per-frequency Chebyshev recurrence in the
// tau-averaged propagator B̄ and the Lanczos eigen-bound estimate.
//
// Reference semantics: ldiv!(u', P, u) src/KPMPreconditioner.jl:355-414 (Sym), :488-550 (Asym);
// calculate_bounds! :625-658; B̄ = Γ̄ D̄ Γ̄ᴴ (Sym) / D̄ Γ̄ (Asym) as in JDQMCFramework's
// Sym/AsymChkbrdPropagator; kpm_lmul!/lanczos! restated from SmoQyKPMCore (three-term Chebyshev
// recurrence on B̄ rescaled to [-1,1]; plain Lanczos) — third-party source absent, see DESIGN.md.
//
// Mapping: after the tau-FFT the device layout v[ω][s][i] already has one frequency of one
// system as a contiguous N-vector, so there is no transpose (the reference needs two,
// :378/:403).  One workgroup owns one (ω, system): the vector sits in LDS, each colour of B̄ is
// one barrier-separated stage with lane == bond, and the recurrence state (T_{k-1}, T_k, the
// running sum) stays in LDS next to it.  Workgroups are issued heaviest expansion order first.
#include "smoqy_internal.h"

namespace smoqy {

__device__ __forceinline__ double wsum_k(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum_real(double v, double *red /* >= 9 doubles */)
{
    v = wsum_k(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < nwave; ++w) t += red[w];
        red[8] = t;
    }
    __syncthreads();
    return red[8];
}

__device__ __forceinline__ void bbar_colour(double2 *W, const int2 *__restrict__ bonds, const double *__restrict__ cbar, const double *__restrict__ sbar, int cb, int ce)
{
    for (int h = cb + (int)threadIdx.x; h < ce; h += (int)blockDim.x) {
        const int2 b = bonds[h];
        const double c = cbar[h], s = sbar[h];
        const double2 a = W[b.x], d = W[b.y];
        W[b.x] = make_double2(c * a.x + s * d.x, c * a.y + s * d.y);
        W[b.y] = make_double2(c * d.x + s * a.x, c * d.y + s * a.y);
    }
    __syncthreads();
}

// MODE 0: Sym B̄ = Γ̄ D̄ Γ̄ᴴ;  1: Asym B̄ = D̄ Γ̄;  2: B̄ᵀB̄ for Asym = Γ̄ᴴ D̄² Γ̄ (KPMPreconditioner.jl:661-679)
template <int MODE>
__device__ __forceinline__ void bbar_apply(double2 *W, int N, int ncol, const int2 *bonds, const int *col_off, const double *dbar, const double *cbar, const double *sbar)
{
    if (MODE == 0)
        for (int c = ncol - 1; c >= 0; --c) bbar_colour(W, bonds, cbar, sbar, col_off[c], col_off[c + 1]);
    else
        for (int c = 0; c < ncol; ++c) bbar_colour(W, bonds, cbar, sbar, col_off[c], col_off[c + 1]);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double d = (MODE == 2) ? dbar[i] * dbar[i] : dbar[i];
        W[i] = make_double2(d * W[i].x, d * W[i].y);
    }
    __syncthreads();
    if (MODE == 0)
        for (int c = 0; c < ncol; ++c) bbar_colour(W, bonds, cbar, sbar, col_off[c], col_off[c + 1]);
    else if (MODE == 2)
        for (int c = ncol - 1; c >= 0; --c) bbar_colour(W, bonds, cbar, sbar, col_off[c], col_off[c + 1]);
}

__device__ __forceinline__ double2 cmulk(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// v <- Σ_k coefs[k] T_k(B') v with B' = (B̄ - avg)/mag  (kpm_lmul!, restated); v lives in ACC on exit
template <int MODE>
__device__ __forceinline__ void kpm_poly(double2 *W, double2 *A1, double2 *A2, double2 *ACC, const double2 *__restrict__ coefs, int n, bool conj_coefs, double avg, double mag, const KpmArgs &k,
                                         const double *dbar, const double *cbar, const double *sbar)
{
    const int N = k.N;
    // on entry the input vector is in ACC
    for (int i = threadIdx.x; i < N; i += blockDim.x) { A1[i] = ACC[i]; W[i] = ACC[i]; }
    __syncthreads();
    bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar);
    double2 c0 = coefs[0], c1 = n > 1 ? coefs[1] : make_double2(0.0, 0.0);
    if (conj_coefs) { c0.y = -c0.y; c1.y = -c1.y; }
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
        bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar);
        double2 ck = coefs[kk];
        if (conj_coefs) ck.y = -ck.y;
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

__global__ void __launch_bounds__(kThreads) cheb_kernel(KpmArgs k)
{
    extern __shared__ double2 lds[];
    const int N = k.N, Lt = k.Lt;
    double2 *W = lds, *A1 = W + N, *A2 = A1 + N, *ACC = A2 + N;
    const int rank = blockIdx.x % Lt, sys = blockIdx.x / Lt;
    const int om = (rank & 1) ? Lt - 1 - (rank >> 1) : (rank >> 1);  // heaviest orders first
    const int w = sys / k.nrhs;
    if (!k.active[w]) return;
    if (k.cg && k.cg[sys].done) return;
    const double *dbar = k.dbar + (size_t)w * N, *cbar = k.cbar + (size_t)w * k.Nh, *sbar = k.sbar + (size_t)w * k.Nh;
    const double emin = k.bounds[2 * w], emax = k.bounds[2 * w + 1];
    const double avg = 0.5 * (emax + emin), mag = 0.5 * (emax - emin);
    double2 *v = k.v + ((size_t)om * k.nsys + sys) * N;
    const int Lo2 = (Lt + 1) / 2;
    if (k.is_sym) {
        const int slot = om >= Lo2 ? Lt - om - 1 : om;  // :387
        const int n = k.order[(size_t)w * k.nslot + slot];
        const double2 *coefs = k.coefs + ((size_t)w * k.nslot + slot) * k.maxorder;
        if (n > 1) {
            for (int i = threadIdx.x; i < N; i += blockDim.x) ACC[i] = v[i];
            __syncthreads();
            kpm_poly<0>(W, A1, A2, ACC, coefs, n, false, avg, mag, k, dbar, cbar, sbar);  // :394
            for (int i = threadIdx.x; i < N; i += blockDim.x) v[i] = ACC[i];
        } else {
            const double c = coefs[0].x;
            for (int i = threadIdx.x; i < N; i += blockDim.x) v[i] = make_double2(c * v[i].x, c * v[i].y);  // :398
        }
    } else {
        const int n = k.order[(size_t)w * k.nslot + om];
        const double2 *coefs = k.coefs + ((size_t)w * k.nslot + om) * k.maxorder;
        if (n > 1) {  // :520-530
            const int omc = Lt - om - 1;
            const double2 *coefs_c = k.coefs + ((size_t)w * k.nslot + omc) * k.maxorder;
            for (int i = threadIdx.x; i < N; i += blockDim.x) ACC[i] = v[i];
            __syncthreads();
            kpm_poly<1>(W, A1, A2, ACC, coefs_c, k.order[(size_t)w * k.nslot + omc], false, avg, mag, k, dbar, cbar, sbar);
            kpm_poly<1>(W, A1, A2, ACC, coefs, n, false, avg, mag, k, dbar, cbar, sbar);
            for (int i = threadIdx.x; i < N; i += blockDim.x) v[i] = ACC[i];
        } else {
            const double c = coefs[0].x * coefs[0].x + coefs[0].y * coefs[0].y;  // :534
            for (int i = threadIdx.x; i < N; i += blockDim.x) v[i] = make_double2(c * v[i].x, c * v[i].y);
        }
    }
}

void launch_cheb(hipStream_t st, const KpmArgs &k)
{
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void *)cheb_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        configured = true;
    }
    const size_t lds = sizeof(double2) * 4 * (size_t)k.N;
    hipLaunchKernelGGL(cheb_kernel, dim3((unsigned)(k.Lt * k.nsys)), dim3(kThreads), lds, st, k);
}

// lanczos! (SmoQyKPMCore, restated): n-step Lanczos on B̄ (Sym) or B̄ᵀB̄ (Asym) from the host
// supplied start vector; one workgroup, everything in LDS.
template <int MODE>
__global__ void __launch_bounds__(kThreads) lanczos_kernel(KpmArgs k, int w, const double *__restrict__ randvec, int nsteps, double *alpha, double *beta)
{
    extern __shared__ double2 lds[];
    __shared__ double red[9];
    const int N = k.N;
    double2 *W = lds, *VK = W + N, *VKM = VK + N;
    const double *dbar = k.dbar + (size_t)w * N, *cbar = k.cbar + (size_t)w * k.Nh, *sbar = k.sbar + (size_t)w * k.Nh;
    double acc = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc += randvec[i] * randvec[i];
    const double nrm = sqrt(block_sum_real(acc, red));
    for (int i = threadIdx.x; i < N; i += blockDim.x) { VK[i] = make_double2(randvec[i] / nrm, 0.0); VKM[i] = make_double2(0.0, 0.0); }
    __syncthreads();
    double bprev = 0.0;
    for (int s = 0; s < nsteps; ++s) {
        for (int i = threadIdx.x; i < N; i += blockDim.x) W[i] = VK[i];
        __syncthreads();
        bbar_apply<MODE>(W, N, k.ncol, k.bonds, k.col_off, dbar, cbar, sbar);
        acc = 0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) acc += VK[i].x * W[i].x;
        const double al = block_sum_real(acc, red);
        acc = 0;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            const double wv = W[i].x - al * VK[i].x - bprev * VKM[i].x;
            W[i].x = wv;
            acc += wv * wv;
        }
        const double nb = sqrt(block_sum_real(acc, red));
        if (threadIdx.x == 0) {
            alpha[s] = al;
            if (s < nsteps - 1) beta[s] = nb;
        }
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            VKM[i] = VK[i];
            VK[i] = make_double2(W[i].x / nb, 0.0);
        }
        bprev = nb;
        __syncthreads();
    }
}

void launch_lanczos(hipStream_t st, const KpmArgs &k, int w, const double *randvec, int nsteps, double *alpha, double *beta, bool use_BtB)
{
    static bool configured = false;
    if (!configured) {
        (void)hipFuncSetAttribute((const void *)lanczos_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        (void)hipFuncSetAttribute((const void *)lanczos_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        configured = true;
    }
    const size_t lds = sizeof(double2) * 3 * (size_t)k.N;
    if (use_BtB) hipLaunchKernelGGL((lanczos_kernel<2>), dim3(1), dim3(kThreads), lds, st, k, w, randvec, nsteps, alpha, beta);
    else hipLaunchKernelGGL((lanczos_kernel<0>), dim3(1), dim3(kThreads), lds, st, k, w, randvec, nsteps, alpha, beta);
}

}  // namespace smoqy
