// Force terms of the pseudofermion action for gfx950 (SURVEY.md §8(f), rank 1 "next" row):
//   mul_νRe∂M∂x!   src/fermion_det_matrix_dervative.jl:2-113 (Sym), :116-186 (Asym)
//   _mul_νReΔτ∂Kc∂x! :189-245,  _mul_νReΔτ∂V∂x! :249-289
//   mul_νRe∂Λ∂x!   src/holstein_shift_matrix.jl:156-201
//
// Like the matvec, the derivative is independent slice by slice once v[l-1] has been shifted in:
// a workgroup owns a tau-chunk of one system, holds |u'> and |v'> in LDS and walks the checkerboard
// colours exactly in the reference's order (lmul on u', ldiv on v', one colour at a time, lane =
// bond).  Every (coupling, slice) writes its value to its own slot of a contribution buffer and a
// second kernel sums the slots that feed one phonon in a fixed order — no atomics, bit-reproducible.
// This runs once per HMC step (not per CG iteration), so it uses the generic memory-resident bond
// tables rather than the register-resident machinery of the matvec.
#include "smoqy_internal.h"

namespace smoqy {

namespace {

__device__ __forceinline__ int wrapf(int l, int Lt) { return l >= Lt ? l - Lt : (l < 0 ? l + Lt : l); }

// one colour: optional SSH accumulation, then u' <- C u', v' <- C⁻¹ v'  (:53-62 / :98-108 / :170-182)
// T = ComplexF64 (shi != nullptr): the factor is [[c, s], [conj(s), c]], its inverse [[c, -s], [-conj(s), c]] (src/checkerboard_matrix_multiply.jl:60-68,
// 132-141), and the SSH derivative ΔτdK is complex: ν Re[conj(u'_j) dK v'_i + conj(u'_i) conj(dK) v'_j] (:225-227).  Real handles take the
// branches they always took.
__device__ __forceinline__ void colour_step(const ForceArgs &a, double2 *UP, double2 *VP, int nk, int l0, int w, int c, const double *ch, const double *sh, const double *shi, bool accumulate, int pass,
                                            double dtau_k, double nu, bool apply_u, bool apply_v, const double2 *F = nullptr, const int2 *BL = nullptr, const int *OL = nullptr)
{
    const int N = a.N, Lt = a.Lt;
    const int2 *bonds_ = BL ? BL : a.bonds;
    const int *coff_ = OL ? OL : a.col_off;
    for (int h = coff_[c] + (int)threadIdx.x; h < coff_[c + 1]; h += (int)blockDim.x) {
        const int2 b = bonds_[h];
        for (int k = 0; k < nk; ++k) {
            const int l = l0 + k;
            double2 *ur = UP + (size_t)k * N, *vr = VP + (size_t)k * N;
            const double2 ui = ur[b.x], uj = ur[b.y], vi = vr[b.x], vj = vr[b.y];
            if (accumulate) {
                const double *xs = a.x + ((size_t)w * Lt + l) * a.Nph;
                for (int q = a.bond_ptr[h]; q < a.bond_ptr[h + 1]; ++q) {
                    const int cpl = a.bond_cpl[q];
                    const int p = a.s_c2p[2 * cpl], pp = a.s_c2p[2 * cpl + 1];
                    const double dx = xs[pp] - xs[p];                                                            // :223
                    const double dK = dtau_k * (a.s_alpha[cpl] + 2 * a.s_alpha2[cpl] * dx + 3 * a.s_alpha3[cpl] * dx * dx + 4 * a.s_alpha4[cpl] * dx * dx * dx);  // :225
                    // ν Re[conj(u'_j) dK v'_i + conj(u'_i) conj(dK) v'_j]                                        // :227
                    double val = nu * dK * ((uj.x * vi.x + uj.y * vi.y) + (ui.x * vj.x + ui.y * vj.y));
                    if (a.s_alpha_im) {  // complex coupling: Re[dK (A + iB)] + Re[conj(dK) (C + iD)] = dK_re (A + C) - dK_im (B - D)
                        const double dKi = dtau_k * (a.s_alpha_im[cpl] + 2 * a.s_alpha2_im[cpl] * dx + 3 * a.s_alpha3_im[cpl] * dx * dx + 4 * a.s_alpha4_im[cpl] * dx * dx * dx);
                        val -= nu * dKi * ((uj.x * vi.y - uj.y * vi.x) - (ui.x * vj.y - ui.y * vj.x));
                    }
                    a.contrib[((size_t)w * Lt + l) * a.Q + a.Nhol + 2 * cpl + pass] = val;
                }
            }
            const double cc = F ? F[(size_t)k * a.Nh + h].x : ch[(size_t)l * a.Nh + h], ss = F ? F[(size_t)k * a.Nh + h].y : sh[(size_t)l * a.Nh + h];
            if (shi) {
                const double ti = shi[(size_t)l * a.Nh + h];  // s = ss + i ti
                if (apply_u) {
                    ur[b.x] = make_double2(cc * ui.x + (ss * uj.x - ti * uj.y), cc * ui.y + (ss * uj.y + ti * uj.x));
                    ur[b.y] = make_double2(cc * uj.x + (ss * ui.x + ti * ui.y), cc * uj.y + (ss * ui.y - ti * ui.x));
                }
                if (apply_v) {
                    vr[b.x] = make_double2(cc * vi.x - (ss * vj.x - ti * vj.y), cc * vi.y - (ss * vj.y + ti * vj.x));
                    vr[b.y] = make_double2(cc * vj.x - (ss * vi.x + ti * vi.y), cc * vj.y - (ss * vi.y - ti * vi.x));
                }
                continue;
            }
            if (apply_u) {
                ur[b.x] = make_double2(cc * ui.x + ss * uj.x, cc * ui.y + ss * uj.y);
                ur[b.y] = make_double2(cc * uj.x + ss * ui.x, cc * uj.y + ss * ui.y);
            }
            if (apply_v) {
                vr[b.x] = make_double2(cc * vi.x - ss * vj.x, cc * vi.y - ss * vj.y);
                vr[b.y] = make_double2(cc * vj.x - ss * vi.x, cc * vj.y - ss * vi.y);
            }
        }
    }
    __syncthreads();
}

// plain colour on one array (building B v)
__device__ __forceinline__ void colour_plain(const ForceArgs &a, double2 *X, int nk, int l0, int c, const double *ch, const double *sh, const double *shi, const double2 *F = nullptr, const int2 *BL = nullptr, const int *OL = nullptr)
{
    const int N = a.N;
    const int2 *bonds_ = BL ? BL : a.bonds;
    const int *coff_ = OL ? OL : a.col_off;  // (LDS copies made by the kernel's prologue: no dependent global load at the head of a pass)
    for (int h = coff_[c] + (int)threadIdx.x; h < coff_[c + 1]; h += (int)blockDim.x) {
        const int2 b = bonds_[h];
        for (int k = 0; k < nk; ++k) {
            // F: the chunk's (cosh, sinh) staged in LDS by the kernel's prologue (all loads in flight at once) instead of a dependent global load per pass
            const double cc = F ? F[(size_t)k * a.Nh + h].x : ch[(size_t)(l0 + k) * a.Nh + h], ss = F ? F[(size_t)k * a.Nh + h].y : sh[(size_t)(l0 + k) * a.Nh + h];
            double2 *r = X + (size_t)k * N;
            const double2 x = r[b.x], y = r[b.y];
            if (shi) {
                const double ti = shi[(size_t)(l0 + k) * a.Nh + h];
                r[b.x] = make_double2(cc * x.x + (ss * y.x - ti * y.y), cc * x.y + (ss * y.y + ti * y.x));
                r[b.y] = make_double2(cc * y.x + (ss * x.x + ti * x.y), cc * y.y + (ss * x.y - ti * x.x));
                continue;
            }
            r[b.x] = make_double2(cc * x.x + ss * y.x, cc * x.y + ss * y.y);
            r[b.y] = make_double2(cc * y.x + ss * x.x, cc * y.y + ss * x.y);
        }
    }
    __syncthreads();
}

// the same colour on TWO arrays in one pass (one barrier, one read of the bond's factors): X and Y both take C_c
__device__ __forceinline__ void colour_plain2(const ForceArgs &a, double2 *X, double2 *Y, int nk, int l0, int c, const double *ch, const double *sh, const double *shi, const double2 *F = nullptr, const int2 *BL = nullptr, const int *OL = nullptr)
{
    const int N = a.N;
    const int2 *bonds_ = BL ? BL : a.bonds;
    const int *coff_ = OL ? OL : a.col_off;  // (LDS copies made by the kernel's prologue: no dependent global load at the head of a pass)
    for (int h = coff_[c] + (int)threadIdx.x; h < coff_[c + 1]; h += (int)blockDim.x) {
        const int2 b = bonds_[h];
        for (int k = 0; k < nk; ++k) {
            // F: the chunk's (cosh, sinh) staged in LDS by the kernel's prologue (all loads in flight at once) instead of a dependent global load per pass
            const double cc = F ? F[(size_t)k * a.Nh + h].x : ch[(size_t)(l0 + k) * a.Nh + h], ss = F ? F[(size_t)k * a.Nh + h].y : sh[(size_t)(l0 + k) * a.Nh + h];
            const double ti = shi ? shi[(size_t)(l0 + k) * a.Nh + h] : 0.0;
            for (int q = 0; q < 2; ++q) {
                double2 *r = (q == 0 ? X : Y) + (size_t)k * N;
                const double2 x = r[b.x], y = r[b.y];
                if (shi) {
                    r[b.x] = make_double2(cc * x.x + (ss * y.x - ti * y.y), cc * x.y + (ss * y.y + ti * y.x));
                    r[b.y] = make_double2(cc * y.x + (ss * x.x + ti * x.y), cc * y.y + (ss * x.y - ti * x.x));
                } else {
                    r[b.x] = make_double2(cc * x.x + ss * y.x, cc * x.y + ss * y.y);
                    r[b.y] = make_double2(cc * y.x + ss * x.x, cc * y.y + ss * x.y);
                }
            }
        }
    }
    __syncthreads();
}

template <bool SYM>
__global__ void __launch_bounds__(kThreads) dmdx_kernel(ForceArgs a)
{
    extern __shared__ double2 lds[];
    const int chunk = blockIdx.x % a.nchunk, sys = blockIdx.x / a.nchunk;
    const int w = sys / a.nrhs, Lt = a.Lt, N = a.N;
    const int l0 = chunk * a.Tc, nk = min(a.Tc, Lt - l0);
    double2 *UP = a.scratch ? a.scratch + (size_t)blockIdx.x * a.scratch_stride : lds, *VP = UP + (size_t)a.Tc * N;
    const double *expV = a.expV + (size_t)w * Lt * N, *ch = a.ch + (size_t)w * Lt * a.Nh, *sh = a.sh + (size_t)w * Lt * a.Nh;
    const double *shi = a.shi ? a.shi + (size_t)w * Lt * a.Nh : nullptr;
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *u = a.u + (size_t)sys * N, *v = a.v + (size_t)sys * N;
    const double nu = -a.nu;  // the reference passes -ν to the helpers (:52, :83, :97, :160, :173)
    // v'[l] = ∓ v[l-1] (+ on the first slice), u'[l] = u[l]     (:27-30, :42 / :140-143, :152)
    for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
        const int k = idx / N, i = idx - k * N, l = l0 + k;
        const double2 x = v[(size_t)wrapf(l - 1, Lt) * sstride + i];
        VP[idx] = (l == 0) ? x : make_double2(-x.x, -x.y);
        UP[idx] = u[(size_t)l * sstride + i];
    }
    // round 4: the chunk's bond factors go to LDS here, with the slices (every load of the prologue in flight together).  Each of the ten to
    // thirteen colour passes below used to start with a dependent global load of its factors: ~4 us per pass, 42 us per launch at 16 walkers
    double2 *F = a.fac_lds ? VP + (size_t)a.Tc * N : nullptr;
    int2 *BL = F ? reinterpret_cast<int2 *>(F + (size_t)a.Tc * a.Nh) : nullptr;   // ... and the bond table and the colour offsets: the loop head of every pass
    int *OL = BL ? reinterpret_cast<int *>(BL + a.Nh) : nullptr;                  // read them from global memory, two dependent loads per pass
    if (F) {
        for (int h = threadIdx.x; h < a.Nh; h += blockDim.x) BL[h] = a.bonds[h];
        if ((int)threadIdx.x <= a.ncol) OL[threadIdx.x] = a.col_off[threadIdx.x];
    }
    if (F)
        for (int idx = threadIdx.x; idx < nk * a.Nh; idx += blockDim.x) {
            const int k = idx / a.Nh, h = idx - k * a.Nh;
            const size_t row = a.cs_slice0 ? (size_t)0 : (size_t)(l0 + k);  // τ-independent hoppings (the host has shown it for every walker): every slice holds slice 0's values; 12 KB per walker, cache resident, instead of Lτ copies
            F[idx] = make_double2(ch[row * a.Nh + h], sh[row * a.Nh + h]);
        }
    __syncthreads();
    if (SYM) {
        // Holstein couplings only: |u'> := Γᵀ|u'> (:66-69) runs the same colours in the same order as the first half of B on |v'> (:33) and
        // neither reads the other: one pass per colour for both arrays (round 4: three barrier phases and their factor loads fewer of
        // thirteen; the arithmetic per array is unchanged)
        const bool merged = a.Nssh == 0;
        for (int c = a.ncol - 1; c >= 0; --c) {
            if (merged) colour_plain2(a, VP, UP, nk, l0, c, ch, sh, shi, F, BL, OL);
            else colour_plain(a, VP, nk, l0, c, ch, sh, shi, F, BL, OL);                                           // :33
        }
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {                              // :36
            const int k = idx / N, i = idx - k * N;
            const double d = expV[(size_t)(l0 + k) * N + i];
            VP[idx] = make_double2(d * VP[idx].x, d * VP[idx].y);
        }
        __syncthreads();
        for (int c = 0; c < a.ncol; ++c) colour_plain(a, VP, nk, l0, c, ch, sh, shi, F, BL, OL);                   // :39
        if (a.Nssh > 0) {
            for (int c = a.ncol - 1; c >= 0; --c) colour_step(a, UP, VP, nk, l0, w, c, ch, sh, shi, true, 0, a.dtau / 2, nu, true, true, F, BL, OL);  // :50-63
        } else {
            // |u'> := Γᵀ|u'> (colours last..first), |v'> := checkerboard_ldiv!(transposed = true) = colours
            // first..last with inverted factors, exactly as the reference does it (:66-74)
            // (|u'> took its colours together with |v'> above)
            for (int c = 0; c < a.ncol; ++c) colour_step(a, UP, VP, nk, l0, w, c, ch, sh, shi, false, 0, 0.0, 0.0, false, true, F, BL, OL);
        }
    } else {
        for (int c = 0; c < a.ncol; ++c) colour_plain(a, VP, nk, l0, c, ch, sh, shi, F, BL, OL);                   // :146
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {                              // :149
            const int k = idx / N, i = idx - k * N;
            const double d = expV[(size_t)(l0 + k) * N + i];
            VP[idx] = make_double2(d * VP[idx].x, d * VP[idx].y);
        }
        __syncthreads();
    }
    // Holstein term  -ν Re[<u'|Δτ ∂V/∂x|v'>]   (:81-84 / :158-161)
    for (int idx = threadIdx.x; idx < nk * a.Nhol; idx += blockDim.x) {
        const int k = idx / a.Nhol, c = idx - k * a.Nhol, l = l0 + k;
        const int p = a.h_c2p[c], i = a.h_c2s[c];
        const double xx = a.x[((size_t)w * Lt + l) * a.Nph + p];
        const double dV = a.dtau * (a.h_alpha[c] + 2 * a.h_alpha2[c] * xx + 3 * a.h_alpha3[c] * xx * xx + 4 * a.h_alpha4[c] * xx * xx * xx);  // :280
        const double2 uu = UP[(size_t)k * N + i], vv = VP[(size_t)k * N + i];
        a.contrib[((size_t)w * Lt + l) * a.Q + c] = nu * dV * (uu.x * vv.x + uu.y * vv.y);           // :282
    }
    if (a.Nssh > 0) {
        // u' *= exp(-ΔτV), v' /= exp(-ΔτV)   (:87, :90 / :168, :170)
        __syncthreads();
        for (int idx = threadIdx.x; idx < nk * N; idx += blockDim.x) {
            const int k = idx / N, i = idx - k * N;
            const double d = expV[(size_t)(l0 + k) * N + i], di = 1.0 / d;
            UP[idx] = make_double2(d * UP[idx].x, d * UP[idx].y);
            VP[idx] = SYM ? make_double2(di * VP[idx].x, di * VP[idx].y) : make_double2(VP[idx].x / d, VP[idx].y / d);
        }
        __syncthreads();
        if (SYM) {
            for (int c = 0; c < a.ncol; ++c) colour_step(a, UP, VP, nk, l0, w, c, ch, sh, shi, true, 1, a.dtau / 2, nu, true, true, F, BL, OL);     // :95-109
        } else {
            for (int c = a.ncol - 1; c >= 0; --c) colour_step(a, UP, VP, nk, l0, w, c, ch, sh, shi, true, 0, a.dtau, nu, true, true, F, BL, OL);    // :172-183
        }
    }
}

// mul_νRe∂Λ∂x!  (src/holstein_shift_matrix.jl:156-201): up = Mᵀ A Ψ, u = Ψ
__global__ void dldx_kernel(ForceArgs a)
{
    const size_t tot = (size_t)a.nsys * a.Lt * a.Nhol;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % a.Nhol);
        const int l = (int)((idx / a.Nhol) % a.Lt), sys = (int)(idx / ((size_t)a.Nhol * a.Lt));
        const int w = sys / a.nrhs;
        double val = 0.0;
        if (a.h_phsym[c]) {
            const int p = a.h_c2p[c], site = a.h_c2s[c];
            const double xx = a.x[((size_t)w * a.Lt + l) * a.Nph + p];
            const double dL = a.dtau * (a.h_alpha[c] + 3 * a.h_alpha3[c] * xx * xx) / 2 * a.lam[((size_t)w * a.Lt + l) * a.N + site];  // :192
            const int lm = (l == 0) ? a.Lt - 1 : l - 1;
            const size_t sstride = (size_t)a.nsys * a.N;
            const double2 x = a.u[(size_t)lm * sstride + (size_t)sys * a.N + site], y = a.v[(size_t)l * sstride + (size_t)sys * a.N + site];
            val = a.nu * dL * (x.x * y.x + x.y * y.y);                                                                                 // :193
        }
        a.contrib[((size_t)w * a.Lt + l) * a.Q + a.Nhol + 2 * a.Nssh + c] = val;
    }
}

// out[w][l][p] += Σ sign · contrib over the slots that feed phonon p, in list order
__global__ void force_reduce_kernel(ForceArgs a, double *out)
{
    const size_t tot = (size_t)a.nw * a.Lt * a.Nph;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot; idx += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx % a.Nph);
        const size_t wl = idx / a.Nph;
        const double *cb = a.contrib + wl * a.Q;
        double acc = 0.0;
        for (int q = a.ph_ptr[p]; q < a.ph_ptr[p + 1]; ++q) acc += a.ph_sign[q] * cb[a.ph_slot[q]];
        out[idx] += acc;
    }
}


// Device-side update!(fermion_path_integral, elph, x, ±1) + update!(fdm, fpi) + update_Λ!.
// SmoQyDQMC's update! is not under /root/reference; the functional forms follow from the derivatives the
// reference takes of them: ∂V_i/∂x_p = α + 2α₂x + 3α₃x² + 4α₄x³ (src/fermion_det_matrix_dervative.jl:281) gives
// V_i = V⁰_i + Σ_c (αx + α₂x² + α₃x³ + α₄x⁴), and ∂K_ji/∂Δx of the same form with K = -t (:233) gives
// t_h = t⁰_h - Σ_c (αΔx + α₂Δx² + α₃Δx³ + α₄Δx⁴), Δx = x[p′] - x[p].  Then FermionDetMatrix.jl:217, :230-231
// and holstein_shift_matrix.jl:11-12, :37 exactly as fields_kernel / lambda_couple_kernel do them.
__global__ void phonon_fields_kernel(ForceArgs a, const double *__restrict__ V0, const double *__restrict__ t0s, double *__restrict__ expV, double *__restrict__ ch, double *__restrict__ sh,
                                     double *__restrict__ lam, double dtau_k, int do_t, const double *__restrict__ t0s_im, double *__restrict__ shi)
{
    const size_t nV = (size_t)a.nw * a.Lt * a.N, nT = do_t ? (size_t)a.nw * a.Lt * a.Nh : 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nV + nT; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nV) {
            const int i = (int)(idx % a.N);
            const size_t wl = idx / a.N;
            const double *x = a.x + wl * a.Nph;
            double V = V0[i], L = (wl % a.Lt == 0) ? 1.0 : -1.0;
            for (int q = a.site_ptr[i]; q < a.site_ptr[i + 1]; ++q) {
                const int c = a.site_cpl[q];
                const double xp = x[a.h_c2p[c]];
                V += (((a.h_alpha4[c] * xp + a.h_alpha3[c]) * xp + a.h_alpha2[c]) * xp + a.h_alpha[c]) * xp;
                if (a.h_phsym[c]) L *= exp(a.dtau * (a.h_alpha[c] * xp + a.h_alpha3[c] * xp * xp * xp) / 2);
            }
            expV[idx] = exp(-a.dtau * V);
            lam[idx] = L;
        } else {
            const size_t j = idx - nV;
            const int n = (int)(j % a.Nh);
            const double *x = a.x + (j / a.Nh) * a.Nph;
            double tt = t0s[n];
            for (int q = a.bond_ptr[n]; q < a.bond_ptr[n + 1]; ++q) {
                const int c = a.bond_cpl[q];
                const double dx = x[a.s_c2p[2 * c + 1]] - x[a.s_c2p[2 * c]];
                tt -= (((a.s_alpha4[c] * dx + a.s_alpha3[c]) * dx + a.s_alpha2[c]) * dx + a.s_alpha[c]) * dx;
            }
            if (shi) {  // T = ComplexF64: complex bare hopping and couplings; sinh = sign(conj t) sinh(Δτ'|t|), as fields_kernel_c
                double ti = t0s_im ? t0s_im[n] : 0.0;
                if (a.s_alpha_im)
                    for (int q = a.bond_ptr[n]; q < a.bond_ptr[n + 1]; ++q) {
                        const int c = a.bond_cpl[q];
                        const double dx = x[a.s_c2p[2 * c + 1]] - x[a.s_c2p[2 * c]];
                        ti -= (((a.s_alpha4_im[c] * dx + a.s_alpha3_im[c]) * dx + a.s_alpha2_im[c]) * dx + a.s_alpha_im[c]) * dx;
                    }
                const double ab = hypot(tt, ti), arg = dtau_k * ab, sn = sinh(arg);
                ch[j] = cosh(arg);
                sh[j] = ab > 0.0 ? tt / ab * sn : 0.0;
                shi[j] = ab > 0.0 ? -ti / ab * sn : 0.0;
                continue;
            }
            const double arg = dtau_k * fabs(tt);
            ch[j] = cosh(arg);
            sh[j] = (tt > 0 ? 1.0 : (tt < 0 ? -1.0 : 0.0)) * sinh(arg);
        }
    }
}

// Lane-owned form of dmdx_kernel for Holstein couplings (round 4; Sym, real hoppings, <= kFdmColours colours, no SSH coupling): the same
// colour passes in the same order — |u'>, |v'> := Γᵀ (colours last..first), |v'> := D|v'>, |v'> := Γ|v'> (first..last),
// |v'> := (Γᵀ)⁻¹|v'> (inverse factors, first..last), src/fermion_det_matrix_dervative.jl:27-39, 66-74, then the Holstein term :81-84 — on the
// tables of fdm_fast_kernel: a lane owns one (padded) bond per colour and keeps its (cosh, sinh) in registers from the prologue on, the
// slice images sit in LDS in position order, the colour-0 passes around D chain in registers (C₀, D, C₀ without an LDS round trip).  Eight
// barrier phases with one LDS read-modify-write each, where the generic kernel's run-time bond loops cost 0.85 us per pass.
template <int NCOL>
__global__ void __launch_bounds__(1024) dmdx_fast_kernel(ForceArgs a, FdmFast ff)
{
    extern __shared__ double2 lds[];
    const int l = blockIdx.x % a.Lt, sys = blockIdx.x / a.Lt;
    const int w = sys / a.nrhs, Lt = a.Lt, N = a.N, j = threadIdx.x;
    double2 *U = lds, *V = lds + N;
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *u = a.u + (size_t)sys * N + (size_t)l * sstride, *v = a.v + (size_t)sys * N + (size_t)(l == 0 ? Lt - 1 : l - 1) * sstride;
    const double *expV = a.expV + ((size_t)w * Lt + l) * N;
    const double2 *csf = ff.csf + ((size_t)w * Lt + (ff.cs_varies[w] ? l : 0)) * ff.ptotal;
    const double nu = -a.nu;
    int2 b[kFdmColours];
    bool on[kFdmColours];
    double2 cs[kFdmColours];
#pragma unroll
    for (int c = 0; c < kFdmColours; ++c) {
        on[c] = false; b[c] = make_int2(0, 0); cs[c] = make_double2(1.0, 0.0);
        if (c < NCOL) {
            const int idx = ff.poff[c] + j;
            if (idx < ff.poff[c + 1]) { on[c] = true; b[c] = ff.pbonds[idx]; cs[c] = csf[idx]; }
        }
    }
    const int2 s0 = on[0] ? ff.psites[ff.poff[0] + j] : make_int2(0, 0);
    const double di = on[0] ? expV[s0.x] : 1.0, dj = on[0] ? expV[s0.y] : 1.0;
    for (int i = j; i < N; i += blockDim.x) {
        const double2 x = v[i];
        V[ff.pos[i]] = (l == 0) ? x : make_double2(-x.x, -x.y);   // v'[l] = ∓ v[l-1]  (:27-30)
        U[ff.pos[i]] = u[i];
    }
    __syncthreads();
    auto fwd = [](double c, double s, double2 x, double2 y, double2 &ox, double2 &oy) {
        ox = make_double2(c * x.x + s * y.x, c * x.y + s * y.y);
        oy = make_double2(c * y.x + s * x.x, c * y.y + s * x.y);
    };
    auto inv = [](double c, double s, double2 x, double2 y, double2 &ox, double2 &oy) {   // checkerboard_ldiv!: (c, −s)
        ox = make_double2(c * x.x - s * y.x, c * x.y - s * y.y);
        oy = make_double2(c * y.x - s * x.x, c * y.y - s * x.y);
    };
    // Γᵀ on both arrays, colours NCOL-1 .. 1
#pragma unroll
    for (int c = kFdmColours - 1; c >= 1; --c)
        if (c < NCOL) {
            if (on[c]) {
                double2 ox, oy;
                fwd(cs[c].x, cs[c].y, U[b[c].x], U[b[c].y], ox, oy); U[b[c].x] = ox; U[b[c].y] = oy;
                fwd(cs[c].x, cs[c].y, V[b[c].x], V[b[c].y], ox, oy); V[b[c].x] = ox; V[b[c].y] = oy;
            }
            __syncthreads();
        }
    // colour 0 (padded to cover every site): U takes C₀; V takes C₀, then D, then the C₀ that opens Γ
    if (on[0]) {
        double2 ox, oy;
        fwd(cs[0].x, cs[0].y, U[b[0].x], U[b[0].y], ox, oy); U[b[0].x] = ox; U[b[0].y] = oy;
        fwd(cs[0].x, cs[0].y, V[b[0].x], V[b[0].y], ox, oy);
        ox = make_double2(di * ox.x, di * ox.y);
        oy = make_double2(dj * oy.x, dj * oy.y);
        double2 px, py;
        fwd(cs[0].x, cs[0].y, ox, oy, px, py); V[b[0].x] = px; V[b[0].y] = py;
    }
    __syncthreads();
    // the rest of Γ on V: colours 1 .. NCOL-1
#pragma unroll
    for (int c = 1; c < kFdmColours; ++c)
        if (c < NCOL) {
            if (on[c]) { double2 ox, oy; fwd(cs[c].x, cs[c].y, V[b[c].x], V[b[c].y], ox, oy); V[b[c].x] = ox; V[b[c].y] = oy; }
            __syncthreads();
        }
    // (Γᵀ)⁻¹ on V: inverse factors, colours 0 .. NCOL-1
#pragma unroll
    for (int c = 0; c < kFdmColours; ++c)
        if (c < NCOL) {
            if (on[c]) { double2 ox, oy; inv(cs[c].x, cs[c].y, V[b[c].x], V[b[c].y], ox, oy); V[b[c].x] = ox; V[b[c].y] = oy; }
            __syncthreads();
        }
    // Holstein term  -ν Re[<u'|Δτ ∂V/∂x|v'>]   (:81-84)
    for (int c = j; c < a.Nhol; c += blockDim.x) {
        const int p = a.h_c2p[c], i = a.h_c2s[c];
        const double xx = a.x[((size_t)w * Lt + l) * a.Nph + p];
        const double dV = a.dtau * (a.h_alpha[c] + 2 * a.h_alpha2[c] * xx + 3 * a.h_alpha3[c] * xx * xx + 4 * a.h_alpha4[c] * xx * xx * xx);  // :280
        const int q = ff.pos[i];
        const double2 uu = U[q], vv = V[q];
        a.contrib[((size_t)w * Lt + l) * a.Q + c] = nu * dV * (uu.x * vv.x + uu.y * vv.y);           // :282
    }
}

}  // namespace

hipError_t configure_force_kernels(const char **what)
{
    hipError_t first = hipSuccess;
    SMOQY_SET_LDS(dmdx_kernel<true>, 160 * 1024 - 256);
    SMOQY_SET_LDS(dmdx_kernel<false>, 160 * 1024 - 256);
    return first;
}

void launch_dmdx(hipStream_t st, const ForceArgs &a, bool sym, const FdmFast *ff)
{
    static const int fast_env = tuning_env(kTuneDmdxFast);  // SMOQY_DMDX_FAST=0: A/B switch
    if (fast_env != 0 && sym && ff && ff->enabled && !ff->csi && !a.shi && !a.scratch && a.Nssh == 0 && a.Tc == 1 && a.ncol >= 1 && a.ncol <= kFdmColours &&
        sizeof(double2) * 2 * (size_t)a.N <= 64 * 1024) {
        const dim3 grid((unsigned)(a.Lt * a.nsys)), block((unsigned)ff->threads);
        const size_t lds = sizeof(double2) * 2 * (size_t)a.N;
        switch (a.ncol) {
            case 1: hipLaunchKernelGGL((dmdx_fast_kernel<1>), grid, block, lds, st, a, *ff); break;
            case 2: hipLaunchKernelGGL((dmdx_fast_kernel<2>), grid, block, lds, st, a, *ff); break;
            case 3: hipLaunchKernelGGL((dmdx_fast_kernel<3>), grid, block, lds, st, a, *ff); break;
            default: hipLaunchKernelGGL((dmdx_fast_kernel<4>), grid, block, lds, st, a, *ff); break;
        }
        return;
    }
    size_t lds = a.scratch ? 0 : sizeof(double2) * 2 * (size_t)a.Tc * a.N;
    ForceArgs b = a;
    const size_t fac = sizeof(double2) * (size_t)a.Tc * a.Nh + sizeof(int2) * (size_t)a.Nh + sizeof(int) * 16;  // factors, bond table, colour offsets
    b.fac_lds = (!a.scratch && !a.shi && a.Nh > 0 && a.ncol <= 15 && lds + fac <= 96 * 1024) ? 1 : 0;   // real hoppings, slices in LDS: the factors join them
    if (b.fac_lds) lds += fac;
    const dim3 grid((unsigned)(a.nchunk * a.nsys));
    if (sym) hipLaunchKernelGGL((dmdx_kernel<true>), grid, dim3(kThreads), lds, st, b);
    else hipLaunchKernelGGL((dmdx_kernel<false>), grid, dim3(kThreads), lds, st, b);
}

void launch_dldx(hipStream_t st, const ForceArgs &a)
{
    const size_t tot = (size_t)a.nsys * a.Lt * a.Nhol;
    if (tot == 0) return;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dldx_kernel, dim3(blocks), dim3(256), 0, st, a);
}

void launch_force_reduce(hipStream_t st, const ForceArgs &a, double *out)
{
    const size_t tot = (size_t)a.nw * a.Lt * a.Nph;
    if (tot == 0) return;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(force_reduce_kernel, dim3(blocks), dim3(256), 0, st, a, out);
}

void launch_phonon_fields(hipStream_t st, const ForceArgs &a, const double *V0, const double *t0s, double *expV, double *ch, double *sh, double *lam, double dtau_k, bool do_t, const double *t0s_im,
                          double *shi)
{
    const size_t tot = (size_t)a.nw * a.Lt * (a.N + (do_t ? a.Nh : 0));
    if (tot == 0) return;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(phonon_fields_kernel, dim3(blocks), dim3(256), 0, st, a, V0, t0s, expV, ch, sh, lam, dtau_k, do_t ? 1 : 0, t0s_im, shi);
}

}  // namespace smoqy
