// GreensEstimator contractions on the device: the stochastic estimate of G(Δ,0) from the
// solved vectors GR = M⁻¹R and the conjugated random vectors, averaged over translations with FFT cross-correlations.
// Reference: measure_GΔ0! src/Measurements/GreensEstimator.jl:179-233, _aperiodic_copyto! :656-671,
// _translational_average! :677-708.  The (D+1)-dimensional transforms themselves are rocFFT plans (api.hip); these
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

}  // namespace smoqy
