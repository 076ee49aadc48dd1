/* Plain-C caller of libsmoqy_hip.so: what the Julia `ccall` shim of INTEGRATION.md does, written in C.
 *
 *   cc -std=c99 -Iinclude examples/c_abi_demo.c -Lsmoqyelphqmc.jl_amd/csrc -lsmoqy_hip -lm -o c_abi_demo
 *
 * Builds a periodic chain (N sites, two checkerboard colours), sets random fields, applies M and Mᵀ through the host
 * entry points, checks the adjoint identity <u, M v> = <Mᵀ u, v>, solves MᵀM x = b with the device CG and checks the
 * residual with an independent MᵀM apply.  Exit code 0 on success.  (SymFermionDetMatrix, src/FermionDetMatrix.jl:22-111.) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "smoqy_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != 0) {                                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, smoqy_last_error(ctx));   \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static double urand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xFFFFFF) / (double)0x1000000 - 0.5; }

int main(void)
{
    enum { N = 32, LT = 24, NH = N };
    smoqy_ctx *ctx = NULL;
    /* colour-sorted neighbour table (2 x Nh, 1-based, column-major): even bonds, then odd bonds */
    int64_t nt[2 * NH], colors[4] = {1, NH / 2, NH / 2 + 1, NH}, perm[NH];
    int h = 0;
    for (int parity = 0; parity < 2; ++parity)
        for (int i = parity; i < N; i += 2, ++h) {
            nt[2 * h] = i + 1;
            nt[2 * h + 1] = (i + 1) % N + 1;
            perm[h] = h + 1; /* the table is handed over already sorted */
        }
    if (smoqy_create(&ctx, LT, N, NH, 2, nt, colors, /*is_sym*/ 1, /*complex T*/ 0, /*walkers*/ 1, /*nrhs*/ 1, /*device*/ -1) != 0) {
        fprintf(stderr, "smoqy_create failed: %s\n", smoqy_last_error(NULL));
        return 2;
    }
    unsigned seed = 12345u;
    double *V = malloc(sizeof(double) * N * LT), *t = malloc(sizeof(double) * NH * LT);
    for (int k = 0; k < N * LT; ++k) V[k] = urand(&seed);
    for (int k = 0; k < NH * LT; ++k) t[k] = 1.0 + 0.2 * urand(&seed);
    CHECK(smoqy_update_from_path_integral(ctx, 0, V, t, perm, 0.05)); /* update!(fdm, fpi) */

    const int n = LT * N;
    double *u = malloc(16 * n), *v = malloc(16 * n), *Mv = malloc(16 * n), *Mtu = malloc(16 * n), *x = malloc(16 * n), *Ax = malloc(16 * n);
    for (int k = 0; k < 2 * n; ++k) { u[k] = urand(&seed); v[k] = urand(&seed); }
    CHECK(smoqy_matvec(ctx, SMOQY_OP_M, Mv, v, 0, 1));   /* mul_M!  */
    CHECK(smoqy_matvec(ctx, SMOQY_OP_MT, Mtu, u, 0, 1)); /* mul_Mt! */
    double lr = 0, li = 0, rr = 0, ri = 0;
    for (int k = 0; k < n; ++k) { /* conj(a) * b */
        lr += u[2 * k] * Mv[2 * k] + u[2 * k + 1] * Mv[2 * k + 1];
        li += u[2 * k] * Mv[2 * k + 1] - u[2 * k + 1] * Mv[2 * k];
        rr += Mtu[2 * k] * v[2 * k] + Mtu[2 * k + 1] * v[2 * k + 1];
        ri += Mtu[2 * k] * v[2 * k + 1] - Mtu[2 * k + 1] * v[2 * k];
    }
    const double adj = hypot(lr - rr, li - ri) / hypot(lr, li);
    printf("adjoint identity |<u,Mv> - <Mtu,v>| / |<u,Mv>| = %.2e\n", adj);

    int iters = 0;
    double eps = 0;
    CHECK(smoqy_cg_solve(ctx, x, v, /*x_is_b*/ 1, 0, 1, 1e-10, 10000, /*preconditioner*/ 0, &iters, &eps)); /* ldiv!(x, fdm, v) */
    CHECK(smoqy_matvec(ctx, SMOQY_OP_MTM, Ax, x, 0, 1));
    double num = 0, den = 0;
    for (int k = 0; k < 2 * n; ++k) { num += (Ax[k] - v[k]) * (Ax[k] - v[k]); den += v[k] * v[k]; }
    printf("CG: %d iterations, eps %.2e, true residual %.2e\n", iters, eps, sqrt(num / den));
    CHECK(smoqy_destroy(ctx));
    free(V); free(t); free(u); free(v); free(Mv); free(Mtu); free(x); free(Ax);
    return (adj < 1e-12 && sqrt(num / den) < 1e-9 && iters > 0) ? 0 : 3;
}
