/* One rank of the reference's one-walker-per-rank model (tutorials/holstein_honeycomb_mpi.jl:60-72) in plain C99, joined to a walker team
 * that ANOTHER process serves (smoqy_team_serve, see examples/walker_team_ranks.py): this program never touches the GPU and owns no handle.
 * It runs the per-walker sweep of the tutorial (:611-684) with its own random stream — two local-move-like updates
 * (sample_pseudofermion_fields! + calculate_fermionic_action!, src/PFFCalculator.jl:56-116) and one hmc_update! whose trajectory runs on the
 * device for all members at once (src/EFAPFFHMCUpdater.jl:102-276) — and prints one JSON line.
 *
 *   gcc -std=c99 -Iinclude examples/team_member_demo.c -Lsmoqyelphqmc.jl_amd/csrc -lsmoqy_member -lm -o team_member_demo   (libsmoqy_member.so: no HIP / rocFFT on this rank)
 *   ./team_member_demo /team-name <walker index> <sweeps> <Nt> <tol>
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "smoqy_hip.h"

static uint64_t rng_state[2];
static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t next_u64(void) /* xoroshiro128+ */
{
    const uint64_t s0 = rng_state[0];
    uint64_t s1 = rng_state[1];
    const uint64_t r = s0 + s1;
    s1 ^= s0;
    rng_state[0] = rotl(s0, 24) ^ s1 ^ (s1 << 16);
    rng_state[1] = rotl(s1, 37);
    return r;
}
static double uniform01(void) { return ((double)(next_u64() >> 11) + 0.5) / 9007199254740992.0; }
static void randn(double *out, size_t n, double scale) /* Box-Muller */
{
    for (size_t i = 0; i < n; i += 2) {
        const double r = sqrt(-2.0 * log(uniform01())) * scale, a = 6.283185307179586 * uniform01();
        out[i] = r * cos(a);
        if (i + 1 < n) out[i + 1] = r * sin(a);
    }
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        const int rc_ = (call);                                                                      \
        if (rc_) {                                                                                   \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, smoqy_member_last_error(me));        \
            return 1; /* the reference's catch block rejects the update; this demo just stops */   \
        }                                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s /team-name walker sweeps Nt tol\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), sweeps = atoi(argv[3]), Nt = atoi(argv[4]);
    const double tol = atof(argv[5]), tol_force = sqrt(tol), drift = 0.02;
    smoqy_member *me = NULL;
    if (smoqy_member_attach(&me, argv[1], w, 60.0)) {
        fprintf(stderr, "attach failed: %s\n", smoqy_member_last_error(NULL));
        return 1;
    }
    int d[4];
    CHECK(smoqy_member_dims(me, d));
    const int Lt = d[0], N = d[1], Nph = d[3];
    const size_t nx = (size_t)Nph * Lt, nR = 2 * (size_t)Lt * N;
    double *x = malloc(nx * sizeof(double)), *x_new = malloc(nx * sizeof(double)), *pi = malloc(nx * sizeof(double)), *P = malloc(nx * sizeof(double));
    double *R = malloc(nR * sizeof(double)), *rv = malloc((size_t)N * (Nt + 1) * sizeof(double));
    if (!x || !x_new || !pi || !P || !R || !rv) return 3;
    CHECK(smoqy_member_fields(me, x)); /* this rank's phonon fields, as the serving rank announced them */
    rng_state[0] = 0x9E3779B97F4A7C15ull * (uint64_t)(w + 1);
    rng_state[1] = 0xD1B54A32D192ED03ull ^ (uint64_t)w;
    long solves = 0, iters = 0;
    double dH_last = 0.0, Sf_last = 0.0;
    for (int s = 0; s < sweeps; ++s) {
        for (int rep = 0; rep < 2; ++rep) { /* reflection / swap stand-ins: propose, evaluate the action, restore */
            double RdotR, Sf, eps;
            int it;
            randn(R, nR, sqrt(0.5)); /* randn!(rng, Φ), src/PFFCalculator.jl:67 */
            CHECK(smoqy_member_sample_phi(me, R, &RdotR));
            randn(pi, nx, 1.0);
            for (size_t i = 0; i < nx; ++i) x[i] += drift * pi[i];
            randn(rv, (size_t)N, 1.0); /* randn!(rng, v), src/KPMPreconditioner.jl:634 */
            CHECK(smoqy_member_pff_step(me, x, rv, tol, 10000, 1, &Sf, &it, &eps, NULL));
            for (size_t i = 0; i < nx; ++i) x[i] -= drift * pi[i];
            solves += 1; iters += it; Sf_last = Sf;
        }
        double H0[3], H1[3];
        int it;
        randn(R, nR, sqrt(0.5));
        randn(P, nx, 1.0);
        randn(rv, (size_t)N * (Nt + 1), 1.0);
        CHECK(smoqy_member_hmc_update(me, x, R, P, rv, Nt, 1.5707963267948966 / Nt, tol_force, tol, 10000, H0, H1, x_new, &it));
        dH_last = (H1[0] + H1[1] + H1[2]) - (H0[0] + H0[1] + H0[2]);
        const int accept = uniform01() < exp(-dH_last); /* the rank's own Metropolis decision, src/EFAPFFHMCUpdater.jl:247-260 */
        CHECK(smoqy_member_hmc_finish(me, accept));
        if (accept)
            for (size_t i = 0; i < nx; ++i) x[i] = x_new[i];
        solves += Nt + 1; iters += it;
    }
    printf("{\"walker\": %d, \"sweeps\": %d, \"solves\": %ld, \"cg_iterations\": %ld, \"last_dH\": %.6e, \"last_action\": %.6e}\n", w, sweeps, solves, iters, dH_last, Sf_last);
    smoqy_member_detach(me);
    free(x); free(x_new); free(pi); free(P); free(R); free(rv);
    return 0;
}
