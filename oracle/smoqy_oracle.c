/*
 * smoqy_oracle.c — CPU restatement of the SmoQyElPhQMC.jl CG / stochastic-trace hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (smoqyelphqmc_amd/, the HIP
 * library) may link, import or call this file.  It is used by tests/, by
 * __graft_entry__.smoke() as the checker, and by bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned".  The reference is pure Julia and cannot run in the
 * build container (no julia, none of its dependencies, no network) and its own tests hold
 * no golden vectors (they are smoke tests on random seeds).  This restatement follows the
 * reference sources pass for pass (citations per function, paths relative to
 * /root/reference) and is validated by the dense-matrix known-answer tests in
 * tests/test_oracle_dense.py that are derived from the reference docstring definitions.
 *
 * Arithmetic that lives in third-party Julia packages whose source is NOT under
 * /root/reference is restated from its published algorithm and the reference call sites:
 *   - SmoQyKPMCore (>=0.1.4): kpm_coefs!, kpm_lmul!, lanczos!  (Chebyshev-Gauss quadrature
 *     with 2n nodes, three-term recurrence, plain Lanczos; no Jackson damping applied)
 *   - JDQMCFramework (>=1.2.5): SymChkbrdPropagator = G D G^H, AsymChkbrdPropagator = D G
 *   - FFTW (>=1.8): unnormalised forward DFT exp(-2 pi i jk/n), 1/n inverse
 * These only influence CG iteration counts, never a converged solution.
 *
 * Layout (identical to the reference): state vectors are Ltau x N column-major complex128,
 * i.e. element (l, i) lives at index l + Ltau*i, tau is the contiguous axis.  Field arrays
 * expV (Ltau x N), cosh/sinh (Ltau x Nh) are column-major doubles.  The neighbour table is
 * 2 x Nh column-major int64, 1-based, already colour sorted (src/FermionDetMatrix.jl:96).
 *
 * Matrix-element type T: Float64 everywhere by default; the *_c entry points restate T = ComplexF64 (complex hoppings:
 * sinh carries the phase sign(conj t), the bond factor is [[c, s], [conj(s), c]], src/checkerboard_matrix_multiply.jl:60-68,
 * src/FermionDetMatrix.jl:224-231) for the operator, the CG and the KPM preconditioner.  The force terms stay real-T only.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

#define IDX(l, i, Lt) ((size_t)(l) + (size_t)(Lt) * (size_t)(i))

/* ------------------------------------------------------------------------------------ */
/* checkerboard_lmul!  — src/checkerboard_matrix_multiply.jl:26-72                       */
/* bonds [h0, h1) (0-based half open), reversed when transposed (lines 45-47)            */
/* ------------------------------------------------------------------------------------ */
/* shi = imaginary part of sinhΔτt (NULL for real T) */
static void chk_apply(cplx *u, int Lt, const int64_t *nt, const double *ch, const double *sh, const double *shi, int reversed, int inverse, int h0, int h1)
{
    int nb = h1 - h0;
    for (int k = 0; k < nb; ++k) {
        int h = reversed ? (h1 - 1 - k) : (h0 + k);
        int i = (int)nt[2 * h] - 1, j = (int)nt[2 * h + 1] - 1;
        cplx *ui = u + IDX(0, i, Lt), *uj = u + IDX(0, j, Lt);
        const double *c = ch + IDX(0, h, Lt), *s = sh + IDX(0, h, Lt), *si = shi ? shi + IDX(0, h, Lt) : NULL;
        for (int l = 0; l < Lt; ++l) { /* lmul :60-68, ldiv :133-141 */
            cplx a = ui[l], b = uj[l];
            cplx sij = si ? s[l] + I * si[l] : s[l];
            if (inverse) sij = -sij;
            ui[l] = c[l] * a + sij * b;
            uj[l] = c[l] * b + conj(sij) * a;
        }
    }
}

void orc_checkerboard_lmul_c(cplx *u, int Lt, int N, const int64_t *nt, const double *ch, const double *sh, const double *shi, int transposed, int h0, int h1)
{
    (void)N;
    chk_apply(u, Lt, nt, ch, sh, shi, transposed, 0, h0, h1); /* reversed when transposed (:45-47) */
}

void orc_checkerboard_lmul(cplx *u, int Lt, int N, const int64_t *nt, const double *ch,
                           const double *sh, int transposed, int h0, int h1)
{
    orc_checkerboard_lmul_c(u, Lt, N, nt, ch, sh, NULL, transposed, h0, h1);
}

/* ------------------------------------------------------------------------------------ */
/* checkerboard_ldiv!  — src/checkerboard_matrix_multiply.jl:98-145                      */
/* inverse factor [[c,-s],[-conj(s),c]] (c^2-|s|^2 = 1), order reversed when NOT transposed (:118-120) */
/* ------------------------------------------------------------------------------------ */
void orc_checkerboard_ldiv_c(cplx *u, int Lt, int N, const int64_t *nt, const double *ch, const double *sh, const double *shi, int transposed, int h0, int h1)
{
    (void)N;
    chk_apply(u, Lt, nt, ch, sh, shi, !transposed, 1, h0, h1);
}

void orc_checkerboard_ldiv(cplx *u, int Lt, int N, const int64_t *nt, const double *ch,
                           const double *sh, int transposed, int h0, int h1)
{
    orc_checkerboard_ldiv_c(u, Lt, N, nt, ch, sh, NULL, transposed, h0, h1);
}

/* ------------------------------------------------------------------------------------ */
/* update!(fdm, fpi) — src/FermionDetMatrix.jl:208-236                                   */
/* V is N x Ltau, t is Nh x Ltau (column-major, as in FermionPathIntegral); perm 1-based  */
/* ------------------------------------------------------------------------------------ */
void orc_update_fields(double *expV, double *ch, double *sh, int Lt, int N, int Nh,
                       const double *V, const double *t, const int64_t *perm, double dtau,
                       int is_sym)
{
    for (int i = 0; i < N; ++i)
        for (int l = 0; l < Lt; ++l) expV[IDX(l, i, Lt)] = exp(-dtau * V[i + (size_t)N * l]); /* :217 */
    double dt2 = is_sym ? dtau / 2 : dtau; /* :220 */
    for (int h = 0; h < Nh; ++h) {
        int hp = (int)perm[h] - 1; /* :224 */
        for (int l = 0; l < Lt; ++l) {
            double tt = t[hp + (size_t)Nh * l];
            double a = dt2 * fabs(tt);
            double sg = (tt > 0) - (tt < 0); /* sign(conj(t)) for real t */
            ch[IDX(l, h, Lt)] = cosh(a);       /* :230 */
            sh[IDX(l, h, Lt)] = sg * sinh(a);  /* :231 */
        }
    }
}

/* T = ComplexF64: t = t_re + i t_im; sinh carries sign(conj(t)) = conj(t)/|t| (:231) */
void orc_update_fields_c(double *expV, double *ch, double *sh, double *shi, int Lt, int N, int Nh, const double *V, const double *t_re, const double *t_im,
                         const int64_t *perm, double dtau, int is_sym)
{
    for (int i = 0; i < N; ++i)
        for (int l = 0; l < Lt; ++l) expV[IDX(l, i, Lt)] = exp(-dtau * V[i + (size_t)N * l]); /* :217 */
    double dt2 = is_sym ? dtau / 2 : dtau; /* :220 */
    for (int h = 0; h < Nh; ++h) {
        int hp = (int)perm[h] - 1; /* :224 */
        for (int l = 0; l < Lt; ++l) {
            cplx tt = t_re[hp + (size_t)Nh * l] + I * t_im[hp + (size_t)Nh * l];
            double ab = cabs(tt), a = dt2 * ab;
            cplx sg = ab > 0 ? conj(tt) / ab : 0.0; /* sign(conj(t′)); Julia's sign(0) = 0 */
            ch[IDX(l, h, Lt)] = cosh(a);               /* :230 */
            sh[IDX(l, h, Lt)] = creal(sg) * sinh(a);   /* :231 */
            shi[IDX(l, h, Lt)] = cimag(sg) * sinh(a);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* the fermion determinant matrix                                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int Lt, N, Nh, is_sym;
    const int64_t *nt;
    const double *expV, *ch, *sh;
    const double *shi; /* imaginary part of sinhΔτt, NULL for T = Float64 */
    cplx *tmp1, *tmp2; /* src/FermionDetMatrix.jl:53-54 */
} orc_fdm;

/* mul_M! Sym: src/FermionDetMatrix.jl:385-427; Asym: :430-466.  out must not alias in. */
void orc_mul_M(const orc_fdm *f, cplx *out, const cplx *in)
{
    int Lt = f->Lt, N = f->N;
    /* circshift!(u', u, (1,0)) :398 / :443 */
    for (int i = 0; i < N; ++i) {
        out[IDX(0, i, Lt)] = in[IDX(Lt - 1, i, Lt)];
        for (int l = 1; l < Lt; ++l) out[IDX(l, i, Lt)] = in[IDX(l - 1, i, Lt)];
    }
    if (f->is_sym) {
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh); /* :401 */
        for (size_t k = 0; k < (size_t)Lt * N; ++k) out[k] *= f->expV[k];    /* :407 */
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, 0, f->Nh); /* :410 */
    } else {
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, 0, f->Nh); /* :446 */
        for (size_t k = 0; k < (size_t)Lt * N; ++k) out[k] *= f->expV[k];    /* :452 */
    }
    for (int i = 0; i < N; ++i) { /* :416-424 / :455-463 */
        out[IDX(0, i, Lt)] = in[IDX(0, i, Lt)] + out[IDX(0, i, Lt)];
        for (int l = 1; l < Lt; ++l) out[IDX(l, i, Lt)] = in[IDX(l, i, Lt)] - out[IDX(l, i, Lt)];
    }
}

/* mul_Mt! Sym: src/FermionDetMatrix.jl:484-525; Asym: :528-563.  out must not alias in. */
void orc_mul_Mt(const orc_fdm *f, cplx *out, const cplx *in)
{
    int Lt = f->Lt, N = f->N;
    size_t V = (size_t)Lt * N;
    if (f->is_sym) {
        memcpy(out, in, V * sizeof(cplx));                                   /* checkerboard_mul! :497 */
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh);
        for (size_t k = 0; k < V; ++k) out[k] *= f->expV[k];                 /* :503 */
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, 0, f->Nh); /* :506 */
    } else {
        for (size_t k = 0; k < V; ++k) out[k] = f->expV[k] * in[k];          /* :541 */
        orc_checkerboard_lmul_c(out, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh); /* :544 */
    }
    for (int i = 0; i < N; ++i) { /* :512-522 / :550-560 */
        cplx last = in[IDX(Lt - 1, i, Lt)] + out[IDX(0, i, Lt)];
        for (int l = 0; l < Lt - 1; ++l) out[IDX(l, i, Lt)] = in[IDX(l, i, Lt)] - out[IDX(l + 1, i, Lt)];
        out[IDX(Lt - 1, i, Lt)] = last;
    }
}

/* mul_MtM! :329-340 (through tmp1); out may alias in */
void orc_mul_MtM(const orc_fdm *f, cplx *out, const cplx *in)
{
    orc_mul_M(f, f->tmp1, in);
    orc_mul_Mt(f, out, f->tmp1);
}

/* mul_MMt! :357-368 */
void orc_mul_MMt(const orc_fdm *f, cplx *out, const cplx *in)
{
    orc_mul_Mt(f, f->tmp1, in);
    orc_mul_M(f, out, f->tmp1);
}

/* ------------------------------------------------------------------------------------ */
/* Holstein shift matrix Lambda — src/holstein_shift_matrix.jl:47-153 (aliasing allowed)  */
/* op: 0 mul_L, 1 ldiv_L, 2 mul_Lt, 3 ldiv_Lt                                             */
/* ------------------------------------------------------------------------------------ */
void orc_lambda_apply(cplx *out, const cplx *in, const double *Lam, int Lt, int N, int op)
{
    for (int n = 0; n < N; ++n) {
        const cplx *v = in + IDX(0, n, Lt);
        cplx *w = out + IDX(0, n, Lt);
        const double *L = Lam + IDX(0, n, Lt);
        if (op == 0) { /* :47-71 */
            cplx v1 = v[0];
            for (int l = 0; l < Lt - 1; ++l) w[l] = L[l + 1] * v[l + 1];
            w[Lt - 1] = L[0] * v1;
        } else if (op == 1) { /* :74-98 */
            cplx vL = v[Lt - 1];
            for (int l = Lt - 1; l >= 1; --l) w[l] = v[l - 1] / L[l];
            w[0] = vL / L[0];
        } else if (op == 2) { /* :102-126 */
            cplx vL = v[Lt - 1];
            for (int l = Lt - 1; l >= 1; --l) w[l] = L[l] * v[l - 1];
            w[0] = L[0] * vL;
        } else { /* :129-153 */
            cplx v1 = v[0];
            for (int l = 0; l < Lt - 1; ++l) w[l] = v[l + 1] / L[l + 1];
            w[Lt - 1] = v1 / L[0];
        }
    }
}

/* update_Lambda! — src/holstein_shift_matrix.jl:2-44.
 * x is Nph x Ltau column-major; ncoup couplings with 1-based phonon / site ids and a flag
 * saying whether that coupling is of the particle-hole symmetric form. */
void orc_update_lambda(double *Lam, int Lt, int N, const double *x, int Nph, double dtau, int ncoup,
                       const int64_t *coupling_to_phonon, const int64_t *coupling_to_site,
                       const double *alpha, const double *alpha3, const int32_t *ph_sym)
{
    for (int n = 0; n < N; ++n) { /* :11-12 */
        Lam[IDX(0, n, Lt)] = 1.0;
        for (int l = 1; l < Lt; ++l) Lam[IDX(l, n, Lt)] = -1.0;
    }
    for (int c = 0; c < ncoup; ++c) {
        if (!ph_sym[c]) continue;
        int p = (int)coupling_to_phonon[c] - 1, site = (int)coupling_to_site[c] - 1;
        for (int l = 0; l < Lt; ++l) { /* :37 */
            double xp = x[p + (size_t)Nph * l];
            Lam[IDX(l, site, Lt)] *= exp(dtau * (alpha[c] * xp + alpha3[c] * xp * xp * xp) / 2);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* complex DFT of arbitrary length (stands in for FFTW plan_fft!/plan_ifft! along dim 1)  */
/* recursive mixed radix decimation in time; prime factors handled by an O(p^2) butterfly */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int n;
    cplx *w;      /* w[k] = exp(-2 pi i k / n) */
    cplx *work;   /* n scratch */
    cplx *theta;  /* theta[l] = exp(-i pi l / n): the FourierTransformer's twiddle, precomputed once as the reference does
                     (src/FourierTransformer.jl:15) */
    cplx *buf;    /* n scratch of the column transform (the reference's plans are in place and allocation free) */
    int nfac, fac[64];
} orc_fft;

static void fft_rec(const orc_fft *p, int n, int stride_in, const cplx *in, cplx *out, int fi, int sign)
{
    if (n == 1) { out[0] = in[0]; return; }
    if (n == 2) { /* leaf butterflies written out: the element-by-element recursion below them costs more than the arithmetic */
        const cplx a = in[0], b = in[stride_in];
        out[0] = a + b; out[1] = a - b;
        return;
    }
    if (n == 4 && p->fac[fi] == 2) {
        const cplx a = in[0], b = in[stride_in], c = in[2 * (size_t)stride_in], d = in[3 * (size_t)stride_in];
        const cplx e0 = a + c, e1 = a - c, o0 = b + d, o1 = (sign > 0 ? I : -I) * (b - d); /* w4 = exp(-/+ i pi/2) */
        out[0] = e0 + o0; out[1] = e1 + o1; out[2] = e0 - o0; out[3] = e1 - o1;
        return;
    }
    int r = p->fac[fi], m = n / r;
    /* r sub-transforms of length m on the decimated inputs */
    for (int q = 0; q < r; ++q) fft_rec(p, m, stride_in * r, in + (size_t)q * stride_in, out + (size_t)q * m, fi + 1, sign);
    int tw = p->n / n; /* twiddle stride in the master table */
    if (r == 2) { /* the common butterfly without the generic loops: (a, b) -> (a + w b, a - w b), w = exp(-/+ 2 pi i k / n) */
        for (int k = 0; k < m; ++k) {
            cplx w = p->w[k * tw];
            if (sign > 0) w = conj(w);
            cplx a = out[k], b = out[m + k] * w;
            out[k] = a + b;
            out[m + k] = a - b;
        }
        return;
    }
    cplx t[r];
    for (int k = 0; k < m; ++k) {
        for (int q = 0; q < r; ++q) {
            int e = (int)(((long long)q * k * tw) % p->n);
            cplx w = p->w[e];
            if (sign > 0) w = conj(w);
            t[q] = out[(size_t)q * m + k] * w;
        }
        for (int s = 0; s < r; ++s) {
            cplx acc = t[0];
            for (int q = 1; q < r; ++q) {
                int e = (int)(((long long)q * s * m * tw) % p->n);
                cplx w = p->w[e];
                if (sign > 0) w = conj(w);
                acc += t[q] * w;
            }
            p->work[s] = acc;
        }
        for (int s = 0; s < r; ++s) out[(size_t)s * m + k] = p->work[s];
    }
}

orc_fft *orc_fft_create(int n)
{
    orc_fft *p = (orc_fft *)calloc(1, sizeof(orc_fft));
    p->n = n;
    p->w = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    p->work = (cplx *)malloc(sizeof(cplx) * (size_t)(n > 64 ? n : 64));
    for (int k = 0; k < n; ++k) p->w[k] = cexp(-2.0 * M_PI * I * (double)k / (double)n);
    p->theta = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    p->buf = (cplx *)malloc(sizeof(cplx) * (size_t)n);
    for (int l = 0; l < n; ++l) p->theta[l] = cexp(-I * M_PI * (double)l / (double)n);
    int m = n;
    for (int f = 2; m > 1;) {
        if (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
        else { ++f; if (f > 61) { /* large prime factor: not needed by any config */ p->fac[p->nfac++] = m; m = 1; } }
    }
    return p;
}

void orc_fft_destroy(orc_fft *p)
{
    if (!p) return;
    free(p->w); free(p->work); free(p->theta); free(p->buf); free(p);
}

/* in-place transform of one contiguous column; sign<0 forward, >0 backward (unnormalised) */
static void fft_col(const orc_fft *p, cplx *x, cplx *buf, int sign)
{
    fft_rec(p, p->n, 1, x, buf, 0, sign);
    memcpy(x, buf, sizeof(cplx) * (size_t)p->n);
}

/* FourierTransformer — src/FourierTransformer.jl:12-21 (theta), :39-50 (lmul!), :53-64 (ldiv!) */
void orc_ft_forward(const orc_fft *p, cplx *u, int Lt, int N)
{
    double isq = 1.0 / sqrt((double)Lt);
    for (int i = 0; i < N; ++i) {
        cplx *c = u + IDX(0, i, Lt);
        for (int l = 0; l < Lt; ++l) c[l] *= p->theta[l] * isq; /* :46, theta from :15 */
        fft_col(p, c, p->buf, -1);                              /* :47 */
    }
}

void orc_ft_inverse(const orc_fft *p, cplx *u, int Lt, int N)
{
    double sq = sqrt((double)Lt) / (double)Lt; /* FFTW's ifft carries the 1/n (:60), then * sqrt(Lt) / theta (:61) */
    for (int i = 0; i < N; ++i) {
        cplx *c = u + IDX(0, i, Lt);
        fft_col(p, c, p->buf, +1);                                      /* :60 */
        for (int l = 0; l < Lt; ++l) c[l] = c[l] * sq * conj(p->theta[l]); /* :61, 1/theta = conj(theta) since |theta| = 1 */
    }
}

/* ------------------------------------------------------------------------------------ */
/* tau-averaged propagator B-bar acting on N-vectors                                      */
/* SymChkbrdPropagator: B = G D G^H; AsymChkbrdPropagator: B = D G (JDQMCFramework)        */
/* src/KPMPreconditioner.jl:242-260, 604-621                                              */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int N, Nh, is_sym;
    const int64_t *nt;
    double *d, *c, *s; /* means over tau */
    double *si;        /* mean of Im sinhΔτt (all zero for T = Float64) */
    int has_si;        /* 0 for T = Float64 */
} orc_bbar;

static void bbar_chk(const orc_bbar *B, cplx *v, int transposed)
{
    for (int k = 0; k < B->Nh; ++k) {
        int h = transposed ? B->Nh - 1 - k : k;
        int i = (int)B->nt[2 * h] - 1, j = (int)B->nt[2 * h + 1] - 1;
        cplx a = v[i], b = v[j];
        if (B->has_si) { /* T = ComplexF64 */
            cplx sij = B->s[h] + I * B->si[h];
            v[i] = B->c[h] * a + sij * b;
            v[j] = B->c[h] * b + conj(sij) * a;
        } else {         /* real hoppings: real-times-complex products, as the reference's Matrix{Float64} fields give */
            v[i] = B->c[h] * a + B->s[h] * b;
            v[j] = B->c[h] * b + B->s[h] * a;
        }
    }
}

/* y = B x (in place) */
static void bbar_mul(const orc_bbar *B, cplx *v)
{
    if (B->is_sym) {
        bbar_chk(B, v, 1);
        for (int i = 0; i < B->N; ++i) v[i] *= B->d[i];
        bbar_chk(B, v, 0);
    } else {
        bbar_chk(B, v, 0);
        for (int i = 0; i < B->N; ++i) v[i] *= B->d[i];
    }
}

/* y = B^T B x (in place) — mul_B̄ᵀB̄!, src/KPMPreconditioner.jl:661-679 */
static void bbar_mul_BtB(const orc_bbar *B, cplx *v)
{
    bbar_chk(B, v, 0);
    for (int i = 0; i < B->N; ++i) v[i] *= B->d[i] * B->d[i];
    bbar_chk(B, v, 1);
}

/* kpm_lmul! (SmoQyKPMCore, restated): v <- sum_k coefs[k] T_k(B') v, B' = (B - avg)/mag.
 * coefficients may be complex (Asym).  tmp is 3N. */
static void kpm_lmul(const orc_bbar *B, const cplx *coefs, int n, cplx *v, double emin, double emax, cplx *tmp)
{
    int N = B->N;
    double avg = 0.5 * (emax + emin), mag = 0.5 * (emax - emin);
    cplx *a1 = tmp, *a2 = tmp + N, *a3 = tmp + 2 * (size_t)N;
    memcpy(a1, v, sizeof(cplx) * (size_t)N);
    memcpy(a2, v, sizeof(cplx) * (size_t)N);
    bbar_mul(B, a2);
    for (int i = 0; i < N; ++i) a2[i] = (a2[i] - avg * a1[i]) / mag;
    for (int i = 0; i < N; ++i) v[i] = coefs[0] * a1[i] + (n > 1 ? coefs[1] * a2[i] : 0.0);
    for (int k = 2; k < n; ++k) {
        memcpy(a3, a2, sizeof(cplx) * (size_t)N);
        bbar_mul(B, a3);
        for (int i = 0; i < N; ++i) {
            a3[i] = 2.0 * (a3[i] - avg * a2[i]) / mag - a1[i];
            v[i] += coefs[k] * a3[i];
        }
        cplx *t = a1; a1 = a2; a2 = a3; a3 = t;
    }
}

/* kpm_coefs! (SmoQyKPMCore, restated): n Chebyshev coefficients of a real scalar function on
 * [emin, emax] by Chebyshev-Gauss quadrature with 2n nodes (buffer size, KPMPreconditioner.jl:749). */
typedef double (*scalar_fn)(double b, double phi);
static void kpm_coefs(double *coefs, int n, scalar_fn f, double phi, double emin, double emax)
{
    int M = 2 * n;
    double avg = 0.5 * (emax + emin), mag = 0.5 * (emax - emin);
    double g[4096];
    for (int j = 0; j < M; ++j) g[j] = f(avg + mag * cos(M_PI * (j + 0.5) / M), phi);
    for (int k = 0; k < n; ++k) {
        double acc = 0;
        for (int j = 0; j < M; ++j) acc += g[j] * cos(M_PI * k * (j + 0.5) / M);
        coefs[k] = (k == 0 ? 1.0 : 2.0) * acc / M;
    }
}

/* src/KPMPreconditioner.jl:800, 804 */
static double f_sym(double b, double phi) { return 1.0 / (b * b - 2 * b * cos(phi) + 1); }
static double f_asym_re(double b, double phi) { return creal(1.0 / (1.0 - cexp(-I * phi) * b)); }
static double f_asym_im(double b, double phi) { return cimag(1.0 / (1.0 - cexp(-I * phi) * b)); }

/* extreme eigenvalues of a symmetric tridiagonal matrix by Sturm bisection */
static int sturm_count(const double *a, const double *b, int n, double x)
{
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0) ++cnt;
    for (int i = 1; i < n; ++i) {
        double den = (fabs(q) < 1e-300) ? (q < 0 ? -1e-300 : 1e-300) : q;
        q = a[i] - x - b[i - 1] * b[i - 1] / den;
        if (q < 0) ++cnt;
    }
    return cnt;
}

void orc_tridiag_extremes(const double *a, const double *b, int n, double *emin, double *emax)
{
    double lo = a[0], hi = a[0];
    for (int i = 0; i < n; ++i) {
        double r = (i > 0 ? fabs(b[i - 1]) : 0) + (i < n - 1 ? fabs(b[i]) : 0);
        if (a[i] - r < lo) lo = a[i] - r;
        if (a[i] + r > hi) hi = a[i] + r;
    }
    double l = lo, h = hi; /* smallest: first x with count >= 1 */
    for (int it = 0; it < 200; ++it) { double m = 0.5 * (l + h); if (sturm_count(a, b, n, m) >= 1) h = m; else l = m; }
    *emin = 0.5 * (l + h);
    l = lo; h = hi; /* largest: first x with count >= n */
    for (int it = 0; it < 200; ++it) { double m = 0.5 * (l + h); if (sturm_count(a, b, n, m) >= n) h = m; else l = m; }
    *emax = 0.5 * (l + h);
}

/* ------------------------------------------------------------------------------------ */
/* KPMPreconditioner — src/KPMPreconditioner.jl:61-99, 132-170 (state), 198-284 (ctor)     */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int Lt, N, Nh, is_sym, active;
    double rbuf, a1, a2;
    int nlanczos;
    orc_bbar B;
    orc_fft *fft;
    double emin, emax;   /* bounds */
    int *order;          /* Sym: cld(Lt,2); Asym: Lt */
    cplx **coefs;
    int ncoef_slots;
    cplx *v;             /* Lt x N */
    cplx *vt;            /* N x Lt */
    cplx *tmp;           /* N x 3 */
    double *lan_a, *lan_b;
} orc_kpm;

orc_kpm *orc_kpm_create(int Lt, int N, int Nh, int is_sym, const int64_t *nt, double rbuf, int nlanczos, double a1, double a2)
{
    orc_kpm *P = (orc_kpm *)calloc(1, sizeof(orc_kpm));
    P->Lt = Lt; P->N = N; P->Nh = Nh; P->is_sym = is_sym; P->active = 0;
    P->rbuf = rbuf; P->nlanczos = nlanczos;
    P->a1 = is_sym ? 2 * a1 : a1; /* :263 */
    P->a2 = a2;
    P->B.N = N; P->B.Nh = Nh; P->B.is_sym = is_sym; P->B.nt = nt;
    P->B.d = (double *)calloc((size_t)N, sizeof(double));
    P->B.c = (double *)calloc((size_t)(Nh > 0 ? Nh : 1), sizeof(double));
    P->B.s = (double *)calloc((size_t)(Nh > 0 ? Nh : 1), sizeof(double));
    P->B.si = (double *)calloc((size_t)(Nh > 0 ? Nh : 1), sizeof(double));
    P->fft = orc_fft_create(Lt);
    P->ncoef_slots = is_sym ? (Lt + 1) / 2 : Lt; /* :254-257, :268-271 */
    P->order = (int *)calloc((size_t)P->ncoef_slots, sizeof(int));
    P->coefs = (cplx **)calloc((size_t)P->ncoef_slots, sizeof(cplx *));
    P->v = (cplx *)calloc((size_t)Lt * N, sizeof(cplx));
    P->vt = (cplx *)calloc((size_t)Lt * N, sizeof(cplx));
    P->tmp = (cplx *)calloc((size_t)N * 3, sizeof(cplx));
    P->lan_a = (double *)calloc((size_t)nlanczos, sizeof(double));
    P->lan_b = (double *)calloc((size_t)nlanczos, sizeof(double));
    return P;
}

void orc_kpm_destroy(orc_kpm *P)
{
    if (!P) return;
    free(P->B.d); free(P->B.c); free(P->B.s); free(P->B.si);
    orc_fft_destroy(P->fft);
    for (int i = 0; i < P->ncoef_slots; ++i) free(P->coefs[i]);
    free(P->order); free(P->coefs); free(P->v); free(P->vt); free(P->tmp); free(P->lan_a); free(P->lan_b);
    free(P);
}

/* update_kpm_expansion_order! :696-731 and update_kpm_expansion_coefs! :734-795 */
static void kpm_update_expansions(orc_kpm *P)
{
    int Lt = P->Lt, Lo2 = (Lt + 1) / 2;
    for (int l = 0; l < P->ncoef_slots; ++l) {
        double phi = 2 * M_PI / Lt * (l + 0.5); /* :220 */
        if (phi > M_PI) phi = 2 * M_PI - phi;   /* :710 */
        int n = (int)floor((P->emax - P->emin) * (P->a1 / phi + P->a2)); /* :711 */
        if (n < 1) n = 1;
        if (n != P->order[l]) {
            P->order[l] = n;
            P->coefs[l] = (cplx *)realloc(P->coefs[l], sizeof(cplx) * (size_t)n);
        }
    }
    double re[2048], im[2048];
    for (int l = 0; l < Lo2; ++l) {
        int n = P->order[l];
        double phi = 2 * M_PI / Lt * (l + 0.5);
        if (P->is_sym) {
            kpm_coefs(re, n, f_sym, phi, P->emin, P->emax); /* :752 */
            for (int k = 0; k < n; ++k) P->coefs[l][k] = re[k];
        } else {
            kpm_coefs(re, n, f_asym_re, phi, P->emin, P->emax); /* :783 */
            kpm_coefs(im, n, f_asym_im, phi, P->emin, P->emax); /* :787 */
            for (int k = 0; k < n; ++k) {
                P->coefs[l][k] = re[k] + I * im[k];
                P->coefs[Lt - l - 1][k] = re[k] - I * im[k]; /* :791 */
            }
        }
    }
}

/* lanczos! (SmoQyKPMCore, restated): plain n-step Lanczos from the start vector v0 (+ i v0i when the matrix element type is
 * complex: randn!(rng, v) on a Vector{ComplexF64} draws complex deviates, src/KPMPreconditioner.jl:634).  The operator is Hermitian,
 * so alpha = Re<v_k, A v_k> and beta = |w| are real. */
static void lanczos(orc_kpm *P, const double *v0, const double *v0i, int use_BtB)
{
    int N = P->N, n = P->nlanczos;
    if (N <= 0 || n <= 0) return;
    cplx *vk = (cplx *)calloc((size_t)N, sizeof(cplx)), *vkm = (cplx *)calloc((size_t)N, sizeof(cplx)), *w = (cplx *)calloc((size_t)N, sizeof(cplx));
    double nrm = 0;
    for (int i = 0; i < N; ++i) nrm += v0[i] * v0[i] + (v0i ? v0i[i] * v0i[i] : 0.0);
    nrm = sqrt(nrm);
    for (int i = 0; i < N; ++i) vk[i] = (v0[i] + (v0i ? I * v0i[i] : 0.0)) / nrm;
    double beta = 0;
    for (int k = 0; k < n; ++k) {
        memcpy(w, vk, sizeof(cplx) * (size_t)N);
        if (use_BtB) bbar_mul_BtB(&P->B, w); else bbar_mul(&P->B, w);
        double alpha = 0;
        for (int i = 0; i < N; ++i) alpha += creal(vk[i]) * creal(w[i]) + cimag(vk[i]) * cimag(w[i]);
        P->lan_a[k] = alpha;
        double nb = 0;
        for (int i = 0; i < N; ++i) { w[i] = w[i] - alpha * vk[i] - beta * vkm[i]; nb += creal(w[i]) * creal(w[i]) + cimag(w[i]) * cimag(w[i]); }
        nb = sqrt(nb);
        if (k < n - 1) P->lan_b[k] = nb;
        beta = nb;
        memcpy(vkm, vk, sizeof(cplx) * (size_t)N);
        for (int i = 0; i < N; ++i) vk[i] = w[i] / nb;
    }
    free(vk); free(vkm); free(w);
}

/* update_preconditioner! — src/KPMPreconditioner.jl:554-597.
 * randvec: the N normal deviates the caller's rng would have produced at :634 / :652. */
void orc_kpm_update_c(orc_kpm *P, const double *expV, const double *ch, const double *sh, const double *shi, const double *randvec, const double *randvec_im)
{
    int Lt = P->Lt, N = P->N, Nh = P->Nh;
    /* update_B̄! :604-621: means over tau */
    for (int i = 0; i < N; ++i) { double a = 0; for (int l = 0; l < Lt; ++l) a += expV[IDX(l, i, Lt)]; P->B.d[i] = a / Lt; }
    for (int h = 0; h < Nh; ++h) {
        double a = 0, b = 0;
        double bi = 0;
        for (int l = 0; l < Lt; ++l) { a += ch[IDX(l, h, Lt)]; b += sh[IDX(l, h, Lt)]; if (shi) bi += shi[IDX(l, h, Lt)]; }
        P->B.c[h] = a / Lt; P->B.s[h] = b / Lt; P->B.si[h] = bi / Lt;
    }
    P->B.has_si = shi != NULL;
    /* calculate_bounds! :625-658 */
    double emin, emax;
    lanczos(P, randvec, randvec_im, !P->is_sym);
    orc_tridiag_extremes(P->lan_a, P->lan_b, P->nlanczos, &emin, &emax);
    if (!P->is_sym) { emin = sqrt(emin); emax = sqrt(emax); } /* :655 */
    emin *= (1 - P->rbuf); /* :569-570 */
    emax *= (1 + P->rbuf);
    if (0.0 < emin && emin < 1.0 && 1.0 < emax && emax < 2.0) { /* :573 */
        P->active = 1;
        if (fabs((emin - P->emin) / P->emin) > P->rbuf / 2 || fabs((emax - P->emax) / P->emax) > P->rbuf / 2) { /* :582 */
            P->emin = emin; P->emax = emax;
            kpm_update_expansions(P);
        }
    } else {
        P->active = 0; /* :593 */
    }
}

void orc_kpm_update(orc_kpm *P, const double *expV, const double *ch, const double *sh, const double *randvec)
{
    orc_kpm_update_c(P, expV, ch, sh, NULL, randvec, NULL);
}

/* ldiv!(u', P, u) complex methods — Sym: src/KPMPreconditioner.jl:355-414; Asym: :488-550 */
void orc_kpm_apply(orc_kpm *P, cplx *out, const cplx *in)
{
    int Lt = P->Lt, N = P->N, Lo2 = (Lt + 1) / 2;
    size_t V = (size_t)Lt * N;
    if (!P->active) { if (out != in) memcpy(out, in, V * sizeof(cplx)); return; } /* :410 */
    memcpy(P->v, in, V * sizeof(cplx));
    orc_ft_forward(P->fft, P->v, Lt, N);                                   /* :375 */
    for (int i = 0; i < N; ++i) for (int l = 0; l < Lt; ++l) P->vt[(size_t)i + (size_t)N * l] = P->v[IDX(l, i, Lt)]; /* :378 */
    for (int n = 0; n < Lt; ++n) {
        cplx *vn = P->vt + (size_t)N * n;
        if (P->is_sym) {
            int np = n >= Lo2 ? Lt - n - 1 : n; /* :387 (0-based) */
            if (P->order[np] > 1) kpm_lmul(&P->B, P->coefs[np], P->order[np], vn, P->emin, P->emax, P->tmp); /* :394 */
            else for (int i = 0; i < N; ++i) vn[i] *= P->coefs[np][0];                                       /* :398 */
        } else {
            if (P->order[n] > 1) { /* :520-530 */
                kpm_lmul(&P->B, P->coefs[Lt - n - 1], P->order[Lt - n - 1], vn, P->emin, P->emax, P->tmp);
                kpm_lmul(&P->B, P->coefs[n], P->order[n], vn, P->emin, P->emax, P->tmp);
            } else {
                double a = creal(P->coefs[n][0]) * creal(P->coefs[n][0]) + cimag(P->coefs[n][0]) * cimag(P->coefs[n][0]);
                for (int i = 0; i < N; ++i) vn[i] *= a; /* :534 */
            }
        }
    }
    for (int i = 0; i < N; ++i) for (int l = 0; l < Lt; ++l) out[IDX(l, i, Lt)] = P->vt[(size_t)i + (size_t)N * l]; /* :403 */
    orc_ft_inverse(P->fft, out, Lt, N);                                    /* :406 */
}

/* ldiv!(u', P, u) REAL-vector methods — Sym: src/KPMPreconditioner.jl:288-352; Asym: :417-485.  Half the frequencies
 * (n = 1..cld(Lt,2)) are evaluated and the other half is filled in as the complex conjugate (:334 / :465), the result is the real
 * part of the back-transform (:344 / :475).  Never reached by CG (its vectors are always complex); restated so that the mirror's
 * real-vector method has an oracle. */
void orc_kpm_apply_real(orc_kpm *P, double *out, const double *in)
{
    int Lt = P->Lt, N = P->N, Lo2 = (Lt + 1) / 2;
    size_t V = (size_t)Lt * N;
    if (!P->active) { if (out != in) memcpy(out, in, V * sizeof(double)); return; } /* :349 / :480 */
    for (size_t k = 0; k < V; ++k) P->v[k] = in[k];                                  /* mul!(v, U, u) :306 / :438 */
    orc_ft_forward(P->fft, P->v, Lt, N);
    for (int i = 0; i < N; ++i) for (int l = 0; l < Lt; ++l) P->vt[(size_t)i + (size_t)N * l] = P->v[IDX(l, i, Lt)]; /* :309 */
    for (int n = 0; n < Lo2; ++n) {                                                  /* :312 / :444 */
        cplx *vn = P->vt + (size_t)N * n, *vm = P->vt + (size_t)N * (Lt - n - 1);
        if (P->is_sym) {
            if (P->order[n] > 1) kpm_lmul(&P->B, P->coefs[n], P->order[n], vn, P->emin, P->emax, P->tmp); /* :324 */
            else for (int i = 0; i < N; ++i) vn[i] *= P->coefs[n][0];                                       /* :328 */
        } else {
            if (P->order[n] > 1) { /* :450-460 */
                kpm_lmul(&P->B, P->coefs[Lt - n - 1], P->order[Lt - n - 1], vn, P->emin, P->emax, P->tmp);
                kpm_lmul(&P->B, P->coefs[n], P->order[n], vn, P->emin, P->emax, P->tmp);
            } else {
                double a = creal(P->coefs[n][0]) * creal(P->coefs[n][0]) + cimag(P->coefs[n][0]) * cimag(P->coefs[n][0]);
                for (int i = 0; i < N; ++i) vn[i] *= a; /* :464 */
            }
        }
        for (int i = 0; i < N; ++i) vm[i] = conj(vn[i]); /* :334 / :468 (for odd Lt the middle column conjugates itself, as in the reference) */
    }
    for (int i = 0; i < N; ++i) for (int l = 0; l < Lt; ++l) P->v[IDX(l, i, Lt)] = P->vt[(size_t)i + (size_t)N * l]; /* :338 */
    orc_ft_inverse(P->fft, P->v, Lt, N);                                             /* :341 */
    for (size_t k = 0; k < V; ++k) out[k] = creal(P->v[k]);                          /* :344 */
}

int orc_kpm_active(const orc_kpm *P) { return P->active; }
void orc_kpm_bounds(const orc_kpm *P, double *b) { b[0] = P->emin; b[1] = P->emax; }
int orc_kpm_order(const orc_kpm *P, int *order) { for (int i = 0; i < P->ncoef_slots; ++i) order[i] = P->order[i]; return P->ncoef_slots; }
void orc_kpm_lanczos(const orc_kpm *P, double *a, double *b) { memcpy(a, P->lan_a, sizeof(double) * (size_t)P->nlanczos); memcpy(b, P->lan_b, sizeof(double) * (size_t)(P->nlanczos - 1)); }
void orc_kpm_coefs(const orc_kpm *P, int slot, cplx *out) { memcpy(out, P->coefs[slot], sizeof(cplx) * (size_t)P->order[slot]); }
void orc_kpm_bbar(const orc_kpm *P, double *d, double *c, double *s) { memcpy(d, P->B.d, sizeof(double) * (size_t)P->N); memcpy(c, P->B.c, sizeof(double) * (size_t)P->Nh); memcpy(s, P->B.s, sizeof(double) * (size_t)P->Nh); }
/* P-bar apply on a single N vector (testing aid): v <- B-bar v */
void orc_kpm_bbar_mul(const orc_kpm *P, cplx *v) { bbar_mul(&P->B, v); }

/* ------------------------------------------------------------------------------------ */
/* cg_solve! — src/IterativeSolvers/ConjugateGradient.jl:93-167 (P = I), :169-249 (P)      */
/* x_is_b: the `x === b` aliasing case (:112-116).  Returns iterations, *eps = final       */
/* relative residual.                                                                     */
/* ------------------------------------------------------------------------------------ */
static cplx zdot(const cplx *a, const cplx *b, size_t n) { cplx s = 0; for (size_t k = 0; k < n; ++k) s += conj(a[k]) * b[k]; return s; }
static double znorm(const cplx *a, size_t n) { double s = 0; for (size_t k = 0; k < n; ++k) s += creal(a[k]) * creal(a[k]) + cimag(a[k]) * cimag(a[k]); return sqrt(s); }

int orc_cg_solve(const orc_fdm *f, orc_kpm *P, cplx *x, const cplx *b, int x_is_b, double tol, int maxiter, double *eps_out)
{
    size_t V = (size_t)f->Lt * f->N;
    cplx *r = (cplx *)malloc(V * sizeof(cplx)), *p = (cplx *)malloc(V * sizeof(cplx)), *z = (cplx *)malloc(V * sizeof(cplx));
    int use_P = (P != NULL);
    double normb = znorm(b, V);
    if (x_is_b) { memcpy(r, b, V * sizeof(cplx)); memset(x, 0, V * sizeof(cplx)); }
    else { orc_mul_MtM(f, r, x); for (size_t k = 0; k < V; ++k) r[k] = b[k] - r[k]; }
    cplx rho;
    if (use_P) { orc_kpm_apply(P, z, r); memcpy(p, z, V * sizeof(cplx)); rho = zdot(r, z, V); }
    else { memcpy(p, r, V * sizeof(cplx)); rho = zdot(r, r, V); }
    double eps = znorm(r, V) / normb;
    int iters = maxiter;
    if (eps < tol) { iters = 0; goto done; }
    for (int it = 1; it <= maxiter; ++it) {
        orc_mul_MtM(f, z, p);
        cplx alpha = rho / zdot(p, z, V);
        for (size_t k = 0; k < V; ++k) x[k] += alpha * p[k];
        for (size_t k = 0; k < V; ++k) r[k] -= alpha * z[k];
        eps = znorm(r, V) / normb;
        if (eps < tol) { iters = it; break; }
        cplx rho_new;
        if (use_P) { orc_kpm_apply(P, z, r); rho_new = zdot(r, z, V); }
        else rho_new = zdot(r, r, V);
        cplx beta = rho_new / rho;
        rho = rho_new;
        const cplx *src = use_P ? z : r;
        for (size_t k = 0; k < V; ++k) p[k] = src[k] + beta * p[k];
    }
done:
    *eps_out = eps;
    free(r); free(p); free(z);
    return iters;
}

/* convenience constructor used from Python */
orc_fdm *orc_fdm_create_c(int Lt, int N, int Nh, int is_sym, const int64_t *nt, const double *expV, const double *ch, const double *sh, const double *shi)
{
    orc_fdm *f = (orc_fdm *)calloc(1, sizeof(orc_fdm));
    f->Lt = Lt; f->N = N; f->Nh = Nh; f->is_sym = is_sym; f->nt = nt; f->expV = expV; f->ch = ch; f->sh = sh; f->shi = shi;
    f->tmp1 = (cplx *)calloc((size_t)Lt * N, sizeof(cplx));
    f->tmp2 = (cplx *)calloc((size_t)Lt * N, sizeof(cplx));
    return f;
}

orc_fdm *orc_fdm_create(int Lt, int N, int Nh, int is_sym, const int64_t *nt, const double *expV, const double *ch, const double *sh)
{
    return orc_fdm_create_c(Lt, N, Nh, is_sym, nt, expV, ch, sh, NULL);
}

void orc_fdm_destroy(orc_fdm *f)
{
    if (!f) return;
    free(f->tmp1); free(f->tmp2); free(f);
}

/* ------------------------------------------------------------------------------------ */
/* force terms — src/fermion_det_matrix_dervative.jl:2-290, src/holstein_shift_matrix.jl:156-201 */
/* (SURVEY.md §8(f) rank 1).  Couplings are passed flattened:                             */
/*   Holstein coupling c: phonon h_c2p[c], site h_c2s[c] (1-based), polynomial α..α4       */
/*   SSH coupling c: phonons s_c2p[2c], s_c2p[2c+1], acting on checkerboard bond s_bond[c]  */
/*     (1-based index into the colour-sorted neighbour table, i.e. n with perm[n] = hopping) */
/* finite_mass[p] = isfinite(M[p]).  out is Nph x Ltau column-major and is accumulated into. */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int Nph;
    const double *x;
    double dtau;
    const int32_t *finite_mass;
    int Nhol;
    const double *ha, *ha2, *ha3, *ha4;
    const int64_t *h_c2p, *h_c2s;
    const int32_t *h_phsym;
    int Nssh;
    const double *sa, *sa2, *sa3, *sa4;
    const int64_t *s_c2p, *s_bond;
    /* T = ComplexF64: imaginary parts of the SSH couplings (ssh_parameters.α::Vector{T}); NULL for real couplings */
    const double *sa_im, *sa2_im, *sa3_im, *sa4_im;
} orc_elph;

/* _mul_νReΔτ∂Kc∂x! — :189-245, for the bonds [h0, h1) of one colour */
static void dKc_dx(const orc_fdm *f, const orc_elph *e, double nu, const cplx *up, const cplx *vp, double dtau_k, int h0, int h1, double *out)
{
    int Lt = f->Lt;
    for (int c = 0; c < e->Nssh; ++c) {
        int n = (int)e->s_bond[c] - 1;
        if (n < h0 || n >= h1) continue;
        int p = (int)e->s_c2p[2 * c] - 1, pp = (int)e->s_c2p[2 * c + 1] - 1;
        int i = (int)f->nt[2 * n] - 1, j = (int)f->nt[2 * n + 1] - 1;
        for (int l = 0; l < Lt; ++l) {
            double dx = e->x[pp + (size_t)e->Nph * l] - e->x[p + (size_t)e->Nph * l];
            cplx dK = dtau_k * (e->sa[c] + 2 * e->sa2[c] * dx + 3 * e->sa3[c] * dx * dx + 4 * e->sa4[c] * dx * dx * dx);   /* :225 */
            if (e->sa_im) dK += I * dtau_k * (e->sa_im[c] + 2 * e->sa2_im[c] * dx + 3 * e->sa3_im[c] * dx * dx + 4 * e->sa4_im[c] * dx * dx * dx);
            double val = nu * creal(conj(up[IDX(l, j, Lt)]) * dK * vp[IDX(l, i, Lt)] + conj(up[IDX(l, i, Lt)]) * conj(dK) * vp[IDX(l, j, Lt)]);  /* :227 */
            if (e->finite_mass[p]) out[p + (size_t)e->Nph * l] -= val;
            if (e->finite_mass[pp]) out[pp + (size_t)e->Nph * l] += val;
        }
    }
}

/* _mul_νReΔτ∂V∂x! — :249-289 */
static void dV_dx(const orc_fdm *f, const orc_elph *e, double nu, const cplx *up, const cplx *vp, double *out)
{
    int Lt = f->Lt;
    for (int c = 0; c < e->Nhol; ++c) {
        int p = (int)e->h_c2p[c] - 1, i = (int)e->h_c2s[c] - 1;
        if (!e->finite_mass[p]) continue;
        for (int l = 0; l < Lt; ++l) {
            double xx = e->x[p + (size_t)e->Nph * l];
            double dV = e->dtau * (e->ha[c] + 2 * e->ha2[c] * xx + 3 * e->ha3[c] * xx * xx + 4 * e->ha4[c] * xx * xx * xx);
            out[p + (size_t)e->Nph * l] += nu * creal(conj(up[IDX(l, i, Lt)]) * dV * vp[IDX(l, i, Lt)]);
        }
    }
}

/* mul_νRe∂M∂x! — Sym :2-113, Asym :116-186.  colors: 2 x ncol, 1-based inclusive bond ranges. */
void orc_mul_dMdx(const orc_fdm *f, const orc_elph *e, const int64_t *colors, int ncol, double nu, const cplx *u, const cplx *v, double *out)
{
    int Lt = f->Lt, N = f->N;
    size_t V = (size_t)Lt * N;
    cplx *vp = f->tmp1, *up = f->tmp2;
    for (int i = 0; i < N; ++i) { /* circshift + sign :27-30 / :140-143 */
        vp[IDX(0, i, Lt)] = v[IDX(Lt - 1, i, Lt)];
        for (int l = 1; l < Lt; ++l) vp[IDX(l, i, Lt)] = -v[IDX(l - 1, i, Lt)];
    }
    if (f->is_sym) {
        orc_checkerboard_lmul_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh);      /* :33 */
        for (size_t k = 0; k < V; ++k) vp[k] *= f->expV[k];                       /* :36 */
        orc_checkerboard_lmul_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, 0, f->Nh);      /* :39 */
        memcpy(up, u, V * sizeof(cplx));                                           /* :42 */
        if (e->Nssh > 0) {
            for (int c = ncol - 1; c >= 0; --c) {                                 /* :50-63 */
                int h0 = (int)colors[2 * c] - 1, h1 = (int)colors[2 * c + 1];
                dKc_dx(f, e, -nu, up, vp, e->dtau / 2, h0, h1, out);
                orc_checkerboard_lmul_c(up, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, h0, h1);
                orc_checkerboard_ldiv_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, h0, h1);
            }
        } else {                                                                   /* :64-75 */
            orc_checkerboard_lmul_c(up, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh);
            orc_checkerboard_ldiv_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, 0, f->Nh);   /* transposed = true, as in the reference */
        }
        if (e->Nhol > 0) dV_dx(f, e, -nu, up, vp, out);                            /* :81-84 */
        for (size_t k = 0; k < V; ++k) { up[k] *= f->expV[k]; vp[k] *= 1.0 / f->expV[k]; } /* :87, :90 */
        if (e->Nssh > 0) {
            for (int c = 0; c < ncol; ++c) {                                      /* :95-109 */
                int h0 = (int)colors[2 * c] - 1, h1 = (int)colors[2 * c + 1];
                dKc_dx(f, e, -nu, up, vp, e->dtau / 2, h0, h1, out);
                orc_checkerboard_lmul_c(up, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, h0, h1);
                orc_checkerboard_ldiv_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, h0, h1);
            }
        }
    } else {
        orc_checkerboard_lmul_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, 0, f->Nh);      /* :146 */
        for (size_t k = 0; k < V; ++k) vp[k] *= f->expV[k];                       /* :149 */
        memcpy(up, u, V * sizeof(cplx));                                           /* :152 */
        if (e->Nhol > 0) dV_dx(f, e, -nu, up, vp, out);                            /* :158-161 */
        if (e->Nssh > 0) {                                                         /* :166-183 */
            for (size_t k = 0; k < V; ++k) { up[k] = f->expV[k] * up[k]; vp[k] = vp[k] / f->expV[k]; }
            for (int c = ncol - 1; c >= 0; --c) {
                int h0 = (int)colors[2 * c] - 1, h1 = (int)colors[2 * c + 1];
                dKc_dx(f, e, -nu, up, vp, e->dtau, h0, h1, out);
                orc_checkerboard_lmul_c(up, Lt, N, f->nt, f->ch, f->sh, f->shi, 0, h0, h1);
                orc_checkerboard_ldiv_c(vp, Lt, N, f->nt, f->ch, f->sh, f->shi, 1, h0, h1);
            }
        }
    }
}

/* mul_νRe∂Λ∂x! — src/holstein_shift_matrix.jl:156-201 */
void orc_mul_dLdx(const orc_elph *e, const double *Lam, int Lt, int N, double nu, const cplx *up, const cplx *u, double *out)
{
    (void)N;
    for (int c = 0; c < e->Nhol; ++c) {
        if (!e->h_phsym[c]) continue;
        int p = (int)e->h_c2p[c] - 1, site = (int)e->h_c2s[c] - 1;
        for (int l = 0; l < Lt; ++l) {
            double xx = e->x[p + (size_t)e->Nph * l];
            double dL = e->dtau * (e->ha[c] + 3 * e->ha3[c] * xx * xx) / 2 * Lam[IDX(l, site, Lt)]; /* :192 */
            int lm = (l == 0) ? Lt - 1 : l - 1;                                                    /* mod1(l-1, Lτ) */
            out[p + (size_t)e->Nph * l] += nu * creal(conj(up[IDX(lm, site, Lt)]) * dL * u[IDX(l, site, Lt)]); /* :193 */
        }
    }
}
