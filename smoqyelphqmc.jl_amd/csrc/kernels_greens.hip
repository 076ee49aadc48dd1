// GreensEstimator contractions on the device: the stochastic estimate of G(Δ,0) from the
// solved vectors GR = M⁻¹R and the conjugated random vectors, averaged over translations with FFT cross-correlations.
// Reference: measure_GΔ0! src/Measurements/GreensEstimator.jl:179-233, _aperiodic_copyto! :656-671,
// _translational_average! :677-708.  The (D+1)-dimensional transforms themselves are rocFFT plans (api_greens.hip); these
// kernels are the data movement around them.
#include "smoqy_internal.h"

namespace smoqy {

namespace {

// A[sys][c][t2], t2 = 0..2Lτ-1 fastest: the orbital-`orb` component of every unit cell c, copied aperiodically along τ
// (second half negated, :656-671); conj = 1 takes the complex conjugate (Rt = conj(R), :171)
__global__ void ge_gather_kernel(const double2 *__restrict__ v, double2 *__restrict__ A, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int conj)
{
    const size_t L2 = 2 * (size_t)Lt, tot = (size_t)nsys * Nc * L2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int t2 = (int)(idx % L2);
        const size_t q = idx / L2;
        const int c = (int)(q % Nc), sys = (int)(q / Nc);
        const int l = t2 < Lt ? t2 : t2 - Lt;
        double2 x = v[((size_t)l * nsys + sys) * N + orb + (size_t)n_orb * c];
        if (conj) x.y = -x.y;
        if (t2 >= Lt) { x.x = -x.x; x.y = -x.y; }
        A[idx] = x;
    }
}

// P[w][k] = Σ_rv Â[w, rv][k] · B̃[w, rv][k]: the element-wise product of :692, summed over the random vectors of a
// walker before the last transform (the transform is linear; the reference transforms every term and then adds)
__global__ void ge_product_kernel(const double2 *__restrict__ Ah, const double2 *__restrict__ Bh, double2 *__restrict__ P, size_t n2, int nrhs, int nw)
{
    const size_t tot = (size_t)nw * n2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t k = idx % n2;
        const int w = (int)(idx / n2);
        double2 acc = make_double2(0.0, 0.0);
        for (int rv = 0; rv < nrhs; ++rv) {
            const size_t off = ((size_t)w * nrhs + rv) * n2 + k;
            const double2 a = Ah[off], b = Bh[off];
            acc.x += a.x * b.x - a.y * b.y;
            acc.y += a.x * b.y + a.y * b.x;
        }
        P[idx] = acc;
    }
}

// out[w][c][τ], τ = 0..Lτ fastest: S on [0, β-Δτ], S[0] copied to τ = β (:697-705), the 1/Nrv of :219 and the two 1/n of
// the normalised inverse transforms folded into `scale`; then G(r,β) = δ(r)δ_ab - G(r,0) (:221-227)
__global__ void ge_finalize_gd0_kernel(const double2 *__restrict__ S, double2 *__restrict__ out, int Lt, int Nc, int nw, double scale, int same_orbital)
{
    const size_t L1 = (size_t)Lt + 1, tot = (size_t)nw * Nc * L1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % L1);
        const size_t q = idx / L1;  // w * Nc + c
        const int c = (int)(q % Nc);
        double2 x = S[q * 2 * (size_t)Lt + (t < Lt ? t : 0)];
        x.x *= scale;
        x.y *= scale;
        if (t == Lt) {
            x.x = -x.x;
            x.y = -x.y;
            if (same_orbital && c == 0) x.x += 1.0;
        }
        out[idx] = x;
    }
}

// four-point estimators (:241-652): S[sys][c][τ] (τ = 0..Lτ-1 fastest) = the orbital-`orb` component of source v
// (conj = 1: Rt = conj(R)) at cell c + r, i.e. ShiftedArrays.circshift(·, (0, -r..., 0)) of :265-268
__global__ void ge_slot_gather_kernel(const double2 *__restrict__ v, double2 *__restrict__ S, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int L1, int L2, int r1, int r2, int conj)
{
    const size_t tot = (size_t)nsys * Nc * Lt;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx % Lt);
        const size_t q = idx / Lt;
        const int c = (int)(q % Nc), sys = (int)(q / Nc);
        const int c1 = c % L1, c2 = c / L1;
        const int s1 = ((c1 + r1) % L1 + L1) % L1, s2 = ((c2 + r2) % L2 + L2) % L2;
        double2 x = v[((size_t)l * nsys + sys) * N + orb + (size_t)n_orb * (s1 + (size_t)L1 * s2)];
        if (conj) x.y = -x.y;
        S[idx] = x;
    }
}

__device__ __forceinline__ double2 cmul_g(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// X[p][k] = S0[i0(p)][k]·S1[i1(p)][k]·[tΔ[k]],  Y[p][k] = S2[i2(p)][k]·S3[i3(p)][k]·[t0[k]]  (_measure_CΔ0! :626-646) for
// every pair p = (n < m) of one walker's random vectors; slot q takes vector m when bit q of `second` is set, n otherwise
__global__ void ge_pair_product_kernel(const double2 *__restrict__ S0, const double2 *__restrict__ S1, const double2 *__restrict__ S2, const double2 *__restrict__ S3, double2 *__restrict__ X,
                                       double2 *__restrict__ Y, const int2 *__restrict__ pairs, int npairs, size_t n1, int second, const double2 *__restrict__ tD, int conj_tD,
                                       const double2 *__restrict__ t0, int conj_t0)
{
    const size_t tot = (size_t)npairs * n1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t k = idx % n1;
        const int2 nm = pairs[idx / n1];
        const int i0 = (second & 1) ? nm.y : nm.x, i1 = (second & 2) ? nm.y : nm.x, i2 = (second & 4) ? nm.y : nm.x, i3 = (second & 8) ? nm.y : nm.x;
        double2 x = cmul_g(S0[(size_t)i0 * n1 + k], S1[(size_t)i1 * n1 + k]);
        double2 y = cmul_g(S2[(size_t)i2 * n1 + k], S3[(size_t)i3 * n1 + k]);
        if (tD) { double2 w = tD[k]; if (conj_tD) w.y = -w.y; x = cmul_g(w, x); }
        if (t0) { double2 w = t0[k]; if (conj_t0) w.y = -w.y; y = cmul_g(w, y); }
        X[idx] = x;
        Y[idx] = y;
    }
}

// P[k] = Σ_p X̂[p][k]·Ỹ[p][k]  (the product of :692 summed over the pairs before the last transform)
__global__ void ge_pair_reduce_kernel(const double2 *__restrict__ X, const double2 *__restrict__ Y, double2 *__restrict__ P, int npairs, size_t n1)
{
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n1; k += (size_t)gridDim.x * blockDim.x) {
        double2 acc = make_double2(0.0, 0.0);
        for (int p = 0; p < npairs; ++p) {
            const double2 t = cmul_g(X[(size_t)p * n1 + k], Y[(size_t)p * n1 + k]);
            acc.x += t.x;
            acc.y += t.y;
        }
        P[k] = acc;
    }
}

// out[c][τ], τ = 0..Lτ: the periodic result on [0, β-Δτ] and its τ = 0 row repeated at τ = β (:697-705), scaled
__global__ void ge_finalize_pairs_kernel(const double2 *__restrict__ S, double2 *__restrict__ out, int Lt, int Nc, double scale)
{
    const size_t L1 = (size_t)Lt + 1, tot = (size_t)Nc * L1;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx % L1);
        const size_t c = idx / L1;
        const double2 x = S[c * (size_t)Lt + (t < Lt ? t : 0)];
        out[idx] = make_double2(x.x * scale, x.y * scale);
    }
}

// scalar boundary terms of the four-point estimators (:325-334, 351-361, 556-566, 586-596): per walker
//   Σ_rv Σ_{τ,c} [bconj(tΔ[τ, c - ts]) · bconj(t0[τ, c])] · GR_og[τ, c - sh, rv] · conj(R_or[τ, c, rv])
// (circshift(a, s)[c] = a[c - s]).  One workgroup per (walker, slab); fixed-order tree reduction, no atomics.
__global__ void ge_boundary_partial_kernel(const double2 *__restrict__ gr, const double2 *__restrict__ r, double2 *__restrict__ part, int Lt, int N, int nsys, int nrhs, int n_orb, int og, int orr, int Nc,
                                           int L1, int L2, int sh1, int sh2, const double2 *__restrict__ tD, int conj_tD, int ts1, int ts2, const double2 *__restrict__ t0, int conj_t0, int nslab)
{
    __shared__ double2 red[256];
    const int w = blockIdx.x / nslab, slab = blockIdx.x % nslab;
    const size_t per = (size_t)nrhs * Nc * Lt;
    double2 acc = make_double2(0.0, 0.0);
    for (size_t idx = (size_t)slab * blockDim.x + threadIdx.x; idx < per; idx += (size_t)nslab * blockDim.x) {
        const int l = (int)(idx % Lt);
        const size_t q = idx / Lt;
        const int c = (int)(q % Nc), sys = w * nrhs + (int)(q / Nc);
        const int c1 = c % L1, c2 = c / L1;
        const int g1 = ((c1 - sh1) % L1 + L1) % L1, g2 = ((c2 - sh2) % L2 + L2) % L2;
        const double2 a = gr[((size_t)l * nsys + sys) * N + og + (size_t)n_orb * (g1 + (size_t)L1 * g2)];
        double2 b = r[((size_t)l * nsys + sys) * N + orr + (size_t)n_orb * c];
        b.y = -b.y;  // Rt = conj(R)
        double2 t = cmul_g(a, b);
        if (tD) {
            const int u1 = ((c1 - ts1) % L1 + L1) % L1, u2 = ((c2 - ts2) % L2 + L2) % L2;
            double2 x = tD[(size_t)(u1 + (size_t)L1 * u2) * Lt + l], y = t0[(size_t)c * Lt + l];
            if (conj_tD) x.y = -x.y;
            if (conj_t0) y.y = -y.y;
            t = cmul_g(cmul_g(x, y), t);
        }
        acc.x += t.x;
        acc.y += t.y;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = blockDim.x / 2; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) { red[threadIdx.x].x += red[threadIdx.x + s2].x; red[threadIdx.x].y += red[threadIdx.x + s2].y; }
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void ge_boundary_final_kernel(const double2 *__restrict__ part, double2 *__restrict__ out, int nslab, double scale)
{
    const int w = blockIdx.x;
    if (threadIdx.x != 0) return;
    double2 t = make_double2(0.0, 0.0);
    for (int k = 0; k < nslab; ++k) { t.x += part[(size_t)w * nslab + k].x; t.y += part[(size_t)w * nslab + k].y; }
    out[w] = make_double2(t.x * scale, t.y * scale);
}

int blocks_for(size_t tot)
{
    size_t b = (tot + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace

void launch_ge_gather(hipStream_t st, const double2 *v, double2 *A, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int conj)
{
    hipLaunchKernelGGL(ge_gather_kernel, dim3(blocks_for((size_t)nsys * Nc * 2 * Lt)), dim3(256), 0, st, v, A, Lt, N, nsys, n_orb, orb, Nc, conj);
}

void launch_ge_product(hipStream_t st, const double2 *Ah, const double2 *Bh, double2 *P, size_t n2, int nrhs, int nw)
{
    hipLaunchKernelGGL(ge_product_kernel, dim3(blocks_for((size_t)nw * n2)), dim3(256), 0, st, Ah, Bh, P, n2, nrhs, nw);
}

void launch_ge_finalize_gd0(hipStream_t st, const double2 *S, double2 *out, int Lt, int Nc, int nw, double scale, int same_orbital)
{
    hipLaunchKernelGGL(ge_finalize_gd0_kernel, dim3(blocks_for((size_t)nw * Nc * (Lt + 1))), dim3(256), 0, st, S, out, Lt, Nc, nw, scale, same_orbital);
}

void launch_ge_slot_gather(hipStream_t st, const double2 *v, double2 *S, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int L1, int L2, int r1, int r2, int conj)
{
    hipLaunchKernelGGL(ge_slot_gather_kernel, dim3(blocks_for((size_t)nsys * Nc * Lt)), dim3(256), 0, st, v, S, Lt, N, nsys, n_orb, orb, Nc, L1, L2, r1, r2, conj);
}

void launch_ge_pair_product(hipStream_t st, const double2 *S0, const double2 *S1, const double2 *S2, const double2 *S3, double2 *X, double2 *Y, const int2 *pairs, int npairs, size_t n1, int second,
                            const double2 *tD, int conj_tD, const double2 *t0, int conj_t0)
{
    hipLaunchKernelGGL(ge_pair_product_kernel, dim3(blocks_for((size_t)npairs * n1)), dim3(256), 0, st, S0, S1, S2, S3, X, Y, pairs, npairs, n1, second, tD, conj_tD, t0, conj_t0);
}

void launch_ge_pair_reduce(hipStream_t st, const double2 *X, const double2 *Y, double2 *P, int npairs, size_t n1)
{
    hipLaunchKernelGGL(ge_pair_reduce_kernel, dim3(blocks_for(n1)), dim3(256), 0, st, X, Y, P, npairs, n1);
}

void launch_ge_boundary(hipStream_t st, const double2 *gr, const double2 *r, double2 *part, double2 *out, int Lt, int N, int nsys, int nrhs, int n_orb, int og, int orr, int Nc, int L1, int L2, int sh1, int sh2,
                        const double2 *tD, int conj_tD, int ts1, int ts2, const double2 *t0, int conj_t0, int nslab, double scale)
{
    const int nw = nsys / nrhs;
    hipLaunchKernelGGL(ge_boundary_partial_kernel, dim3((unsigned)(nw * nslab)), dim3(256), 0, st, gr, r, part, Lt, N, nsys, nrhs, n_orb, og, orr, Nc, L1, L2, sh1, sh2, tD, conj_tD, ts1, ts2, t0, conj_t0, nslab);
    hipLaunchKernelGGL(ge_boundary_final_kernel, dim3((unsigned)nw), dim3(64), 0, st, part, out, nslab, scale);
}

void launch_ge_finalize_pairs(hipStream_t st, const double2 *S, double2 *out, int Lt, int Nc, double scale)
{
    hipLaunchKernelGGL(ge_finalize_pairs_kernel, dim3(blocks_for((size_t)Nc * (Lt + 1))), dim3(256), 0, st, S, out, Lt, Nc, scale);
}

}  // namespace smoqy
