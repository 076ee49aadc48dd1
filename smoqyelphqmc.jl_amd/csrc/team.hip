// Walker teams (include/smoqy_hip.h, "walker teams"): the reference's one-walker-per-rank control flow on ONE batched handle.
//
// The reference runs every Monte Carlo walker as its own MPI rank with its own single-walker FermionDetMatrix / PFFCalculator
// (tutorials/holstein_honeycomb_mpi.jl:60-72).  On one GPU that model does not scale: K processes time-slice the device (measured:
// 6 ranks deliver fewer sweeps/s than 1), K threads with one single-walker handle each top out at four hardware queues.  What does scale
// is ONE handle that carries K walkers and launches every kernel once for all of them — but that needs the K control flows to arrive
// together.  A team is that rendezvous: K host threads (Julia tasks, one per replica) each run the UNCHANGED per-walker update sequence
// and call the team entry points with their own walker index; a call blocks until all K members have made the same call, the last one
// to arrive executes the batched library call for everybody, and each member returns with its own results.  Built on the public C ABI
// only; no kernel knows about teams.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "../../include/smoqy_hip.h"
#include "team_shm.h"

using namespace smoqy_team_detail;


struct Served;

struct smoqy_team {
    smoqy_ctx *c = nullptr;
    int K = 0, Lt = 0, N = 0, Nph = 0;
    int phi = -1, psi = -1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, op = OP_NONE;
    bool running = false;  // the last arrival is inside run_round: the round's outputs go through every member's slot, nobody may leave
    unsigned long gen = 0;
    double timeout_s = 600.0;
    std::vector<Slot> slot;
    std::vector<char> x_seen;  // member w has supplied its phonon fields at least once (later NULLs mean "unchanged")
    bool hmc_pending = false;  // an hmc_update round is waiting for the members' accept / reject decisions
    // page-locked staging in the batched layouts of the C ABI
    void *h_R = nullptr;                                     // Ltau x N x K complex128
    double *h_x = nullptr, *h_rv = nullptr, *h_dS = nullptr;  // Nph x Ltau x K, N x K, Nph x Ltau x K
    double *h_P = nullptr, *h_rvs = nullptr;                  // Nph x Ltau x K, N x K x (kMaxNt + 1)
    // GreensEstimator (smoqy_team_ge_config): follower handle with Nrv systems per walker, its vectors R / GR / MᵀR, staging
    smoqy_ctx *ge = nullptr;
    int ge_Nrv = 0, ge_r = -1, ge_gr = -1, ge_mtr = -1;
    size_t ge_gbytes = 0;
    char *h_GR = nullptr, *h_G = nullptr;
    std::vector<int> ge_it;
    std::vector<double> ge_eps;
    Stage stage() const
    {
        Stage g;
        g.R = (char *)h_R; g.x = h_x; g.rv = h_rv; g.dS = h_dS; g.P = h_P; g.rvs = h_rvs;
        g.GR = h_GR; g.G = h_G; g.Nrv = ge_Nrv; g.gbytes = ge_gbytes;
        return g;
    }
    std::vector<double> Sf, eps, dot, e0, e1;
    std::vector<int> flags;
    std::vector<int> iters;
    std::string err;
    struct Served *served = nullptr;  // published for members of other processes (smoqy_team_serve)
};

static std::string g_team_error;

extern "C" {

const char *smoqy_team_last_error(const smoqy_team *t) { return t ? t->err.c_str() : g_team_error.c_str(); }

int smoqy_team_unserve(smoqy_team *t);

int smoqy_team_destroy(smoqy_team *t)
{
    if (!t) return 0;
    smoqy_team_unserve(t);
    if (t->c) {
        if (t->h_R) smoqy_host_free(t->c, t->h_R);
        if (t->h_x) smoqy_host_free(t->c, t->h_x);
        if (t->h_rv) smoqy_host_free(t->c, t->h_rv);
        if (t->h_dS) smoqy_host_free(t->c, t->h_dS);
        if (t->h_P) smoqy_host_free(t->c, t->h_P);
        if (t->h_rvs) smoqy_host_free(t->c, t->h_rvs);
        if (t->h_GR) smoqy_host_free(t->c, t->h_GR);
        if (t->h_G) smoqy_host_free(t->c, t->h_G);
        if (t->ge) smoqy_destroy(t->ge);
        if (t->phi >= 0) smoqy_vec_free(t->c, t->phi);
        if (t->psi >= 0) smoqy_vec_free(t->c, t->psi);
    }
    delete t;
    return 0;
}

int smoqy_team_create(smoqy_team **out, smoqy_ctx *ctx, int Nph)
{
    if (!out || !ctx || Nph < 0) { g_team_error = "smoqy_team_create: null handle or negative Nph"; return 1; }
    *out = nullptr;
    int d[6];
    if (int rc = smoqy_dims(ctx, d)) return rc;
    if (d[5] != 1) { g_team_error = "smoqy_team_create: the handle must have nrhs = 1 (one system per walker)"; return 1; }
    int tr[8];
    if (int rc = smoqy_traits(ctx, tr)) return rc;
    // the staging buffers and the shared-memory layout hold N doubles per Lanczos start vector; a handle with T = ComplexF64 reads 2N
    // (randn! on a Vector{ComplexF64}, src/KPMPreconditioner.jl:229, 634).  No shipped script has complex hoppings: refused, not mis-staged.
    if (tr[1]) { g_team_error = "smoqy_team_create: handles with complex hoppings (is_complex_T) are not supported by walker teams; drive them through the batched entry points"; return 1; }
    smoqy_team *t = new smoqy_team();
    t->c = ctx; t->Lt = d[0]; t->N = d[1]; t->K = d[4]; t->Nph = Nph;
    t->slot.resize((size_t)t->K);
    t->x_seen.assign((size_t)t->K, 0);
    t->Sf.resize((size_t)t->K); t->eps.resize((size_t)t->K); t->iters.resize((size_t)t->K); t->dot.resize(2 * (size_t)t->K);
    t->e0.resize((size_t)t->K); t->e1.resize((size_t)t->K); t->flags.resize((size_t)t->K);
    const size_t nR = (size_t)t->Lt * t->N * t->K * 16, nx = (size_t)std::max(Nph, 1) * t->Lt * t->K * sizeof(double);
    int rc = smoqy_host_alloc(ctx, &t->h_R, nR);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_x, nx);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_dS, nx);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_rv, (size_t)t->N * t->K * sizeof(double));
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_P, nx);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_rvs, (size_t)t->N * t->K * (kMaxNt + 1) * sizeof(double));
    if (!rc) rc = smoqy_vec_alloc(ctx, &t->phi);
    if (!rc) rc = smoqy_vec_alloc(ctx, &t->psi);
    if (rc) {
        g_team_error = std::string("smoqy_team_create: ") + smoqy_last_error(ctx);
        smoqy_team_destroy(t);
        return rc;
    }
    std::memset(t->h_x, 0, nx);
    *out = t;
    return 0;
}

int smoqy_team_size(const smoqy_team *t, int *K)
{
    if (!t || !K) return 1;
    *K = t->K;
    return 0;
}

int smoqy_team_set_timeout(smoqy_team *t, double seconds)
{
    if (!t || !(seconds > 0)) return 1;
    std::lock_guard<std::mutex> lk(t->m);
    t->timeout_s = seconds;
    return 0;
}

int smoqy_team_vectors(const smoqy_team *t, int *phi, int *psi)
{
    if (!t) return 1;
    if (phi) *phi = t->phi;
    if (psi) *psi = t->psi;
    return 0;
}

}  // extern "C"

static int rendezvous_locked(smoqy_team *t, int w, int op, const Slot &args);

// the batched call of one round, run by the last member to arrive (all the others are blocked in rendezvous)
static int run_round(smoqy_team *t)
{
    smoqy_ctx *c = t->c;
    const int K = t->K;
    if (t->op == OP_SAMPLE && t->hmc_pending) { t->err = "the members owe their decisions first: smoqy_team_hmc_finish must follow smoqy_team_hmc_update"; return 1; }
    if (t->op == OP_SAMPLE) {
        // sample_pseudofermion_fields! (src/PFFCalculator.jl:56-76): Φ = Λᵀ Mᵀ R, |R|² per member
        if (int rc = smoqy_vec_upload(c, t->phi, t->h_R, 0, K)) return rc;
        if (int rc = smoqy_vec_dot(c, t->phi, t->phi, t->dot.data())) return rc;
        if (int rc = smoqy_matvec_v(c, SMOQY_OP_MT, t->phi, t->phi)) return rc;                // lmul_Mt! (:71)
        if (int rc = smoqy_lambda_apply_v(c, SMOQY_LAMBDA_MULT, t->phi, t->phi)) return rc;    // mul_Λᵀ! (:73)
        for (int w = 0; w < K; ++w)
            if (t->slot[w].RdotR) *t->slot[w].RdotR = t->dot[2 * (size_t)w];
        return 0;
    }
    if (t->op == OP_GE_UPDATE || t->op == OP_GE_GD0) {
        if (!t->ge) { t->err = "smoqy_team_ge_*: call smoqy_team_ge_config first"; return 1; }
        if (t->hmc_pending) { t->err = "the members owe their decisions first: smoqy_team_hmc_finish must follow smoqy_team_hmc_update"; return 1; }
        auto fail = [&](int rc) { t->err = smoqy_last_error(t->ge); return rc; };
        const Slot &s0 = t->slot[0];
        if (t->op == OP_GE_UPDATE) {
            // update_greens_estimator! (src/Measurements/GreensEstimator.jl:125-175) of all K members: the follower handle takes the walkers'
            // current fields, update_preconditioner! (:150), MᵀR (:157), K·Nrv solves in ONE batched CG (:159-165)
            for (int w = 0; w < K; ++w) {
                const Slot &s = t->slot[w];
                if (!s.Rrv || !s.rv) { t->err = "smoqy_team_ge_update: random vectors and a Lanczos start vector are needed from every member"; return 1; }
                if (s.tol != s0.tol || s.maxiter != s0.maxiter) { t->err = "smoqy_team_ge_update: the members of a round must pass the same tol / maxiter"; return 1; }
                if (int rc = smoqy_copy_fields(t->ge, w, c, w)) return fail(rc);
            }
            if (int rc = smoqy_precond_update_all(t->ge, t->h_rv)) return fail(rc);
            if (int rc = smoqy_vec_upload(t->ge, t->ge_r, t->h_GR, 0, K * t->ge_Nrv)) return fail(rc);
            if (int rc = smoqy_matvec_v(t->ge, SMOQY_OP_MT, t->ge_mtr, t->ge_r)) return fail(rc);
            if (int rc = smoqy_cg_solve_v(t->ge, t->ge_gr, t->ge_mtr, s0.tol, s0.maxiter, 1, t->ge_it.data(), t->ge_eps.data())) return fail(rc);
            for (int w = 0; w < K; ++w) {
                const Slot &s = t->slot[w];
                int sum = 0;
                double worst = 0.0;
                for (int j = 0; j < t->ge_Nrv; ++j) { sum += t->ge_it[(size_t)w * t->ge_Nrv + j]; worst = std::max(worst, t->ge_eps[(size_t)w * t->ge_Nrv + j]); }
                if (s.iters) *s.iters = sum;
                if (s.eps) *s.eps = worst;
            }
            return 0;
        }
        // measure_GΔ0! (:179-233) for the orbital pair all members ask for: one batched contraction, each member takes its own array
        for (int w = 0; w < K; ++w)
            if (t->slot[w].orb_a != s0.orb_a || t->slot[w].orb_b != s0.orb_b) { t->err = "smoqy_team_ge_measure_GD0: the members of a round must ask for the same orbital pair"; return 1; }
        if (int rc = smoqy_ge_measure_GD0(t->ge, t->ge_gr, t->ge_r, s0.orb_a, s0.orb_b, t->h_G)) return fail(rc);
        return 0;
    }
    if (t->op == OP_FINISH) {
        // each member's Metropolis decision (src/EFAPFFHMCUpdater.jl:252-275): the rejecting ones get their checkpointed fields back
        if (!t->hmc_pending) { t->err = "smoqy_team_hmc_finish without a preceding smoqy_team_hmc_update"; return 1; }
        const size_t nx = (size_t)(t->Nph > 0 ? t->Nph : 1) * t->Lt;
        for (int w = 0; w < K; ++w) {
            t->flags[w] = t->slot[w].accept ? 0 : 1;
            // the staged copy of an accepting member's fields follows the device (a later batched upload sends the staging of ALL members)
            if (t->slot[w].accept) std::memcpy(t->h_x + (size_t)w * nx, t->h_dS + (size_t)w * nx, nx * sizeof(double));
        }
        t->hmc_pending = false;
        if (int rc = smoqy_efa_restore_walkers(c, t->flags.data())) { t->err = smoqy_last_error(c); return rc; }
        return 0;
    }
    if (t->hmc_pending) { t->err = "the members owe their decisions first: smoqy_team_hmc_finish must follow smoqy_team_hmc_update"; return 1; }
    if (t->op == OP_HMC) {
        // hmc_update! (src/EFAPFFHMCUpdater.jl:102-276) of all K members with the trajectory on the device: fields from the members' x,
        // Φ = Λᵀ Mᵀ R (:133), copyto!(x0, x) (:130), momenta from the members' deviates (:142), S_b (:136), the leapfrog (:148-206) in one
        // library call, the final action at `tol` (:217), K and S_b afterwards (:238-244), the proposed fields back to the members
        const Slot &s0 = t->slot[0];
        bool any_x = false;
        for (int w = 0; w < K; ++w) {
            const Slot &s = t->slot[w];
            if (s.x) { t->x_seen[w] = 1; any_x = true; }
            if (!s.R || !s.P || !s.rvs) { t->err = "smoqy_team_hmc_update: R, P and randvecs are needed from every member"; return 1; }
            if (s.Nt != s0.Nt || s.dt != s0.dt || s.tol != s0.tol || s.tol_force != s0.tol_force || s.maxiter != s0.maxiter) {
                t->err = "smoqy_team_hmc_update: the members of a round must pass the same Nt / dt / tolerances / maxiter";
                return 1;
            }
        }
        if (s0.Nt < 1 || s0.Nt > kMaxNt) { t->err = "smoqy_team_hmc_update: Nt outside 1 … 64"; return 1; }
        if (any_x)
            for (int w = 0; w < K; ++w)
                if (!t->x_seen[w]) { t->err = "smoqy_team_hmc_update: a member passed x = NULL before it ever supplied its phonon fields"; return 1; }
        auto fail = [&](int rc) { t->err = smoqy_last_error(c); return rc; };
        if (any_x) if (int rc = smoqy_efa_set_state(c, t->h_x, nullptr)) return fail(rc);
        if (int rc = smoqy_vec_upload(c, t->phi, t->h_R, 0, K)) return fail(rc);
        if (int rc = smoqy_vec_dot(c, t->phi, t->phi, t->dot.data())) return fail(rc);             // S_f at the start = |R|²
        if (int rc = smoqy_matvec_v(c, SMOQY_OP_MT, t->phi, t->phi)) return fail(rc);
        if (int rc = smoqy_lambda_apply_v(c, SMOQY_LAMBDA_MULT, t->phi, t->phi)) return fail(rc);
        if (int rc = smoqy_efa_checkpoint(c, 0)) return fail(rc);
        if (int rc = smoqy_efa_initialize_momentum(c, t->h_P, t->e0.data())) return fail(rc);       // K before
        if (int rc = smoqy_efa_energies(c, nullptr, t->e1.data())) return fail(rc);                // S_b before
        for (int w = 0; w < K; ++w)
            if (double *H = t->slot[w].H0) { H[0] = t->dot[2 * (size_t)w]; H[1] = t->e1[w]; H[2] = t->e0[w]; }
        std::vector<int> its((size_t)K * s0.Nt);
        int rc = smoqy_hmc_trajectory_v(c, t->phi, t->psi, s0.Nt, s0.dt, s0.tol_force, s0.maxiter, 1, t->h_rvs, nullptr, its.data(), nullptr);
        if (!rc) rc = smoqy_pff_step_v(c, t->phi, t->psi, nullptr, t->h_rvs + (size_t)s0.Nt * K * t->N, s0.tol, s0.maxiter, 1, t->Sf.data(), t->iters.data(), t->eps.data(), nullptr);
        if (rc) {
            // the reference's catch block (:176-187): the update is rejected for everybody, the fields go back to the checkpoint
            t->err = smoqy_last_error(c);
            (void)smoqy_efa_checkpoint(c, 1);
            return rc;
        }
        if (int rc2 = smoqy_efa_energies(c, t->e0.data(), t->e1.data())) return fail(rc2);         // K, S_b after
        if (int rc2 = smoqy_efa_get_state(c, t->h_dS, nullptr)) return fail(rc2);                  // the proposed fields, through the force staging
        for (int w = 0; w < K; ++w) {
            const Slot &s = t->slot[w];
            if (s.H1) { s.H1[0] = t->Sf[w]; s.H1[1] = t->e1[w]; s.H1[2] = t->e0[w]; }
            if (s.eps) *s.eps = t->eps[w];
            if (s.iters) {
                int sum = t->iters[w];
                for (int q = 0; q < s0.Nt; ++q) sum += its[(size_t)q * K + w];
                *s.iters = sum;
            }
        }
        t->hmc_pending = true;
        return 0;
    }
    // calculate_fermionic_action! / calculate_derivative_fermionic_action! (src/PFFCalculator.jl:79-157) behind the field update of
    // the caller's move (src/EFAPFFHMCUpdater.jl:200-205, src/reflection_update.jl:99)
    bool any_x = false, any_force = false, all_rv = true;
    for (int w = 0; w < K; ++w) {
        const Slot &s = t->slot[w];
        if (s.x) { t->x_seen[w] = 1; any_x = true; }
        if (s.dSdx) any_force = true;
        if (!s.rv) all_rv = false;
        if (s.tol != t->slot[0].tol || s.maxiter != t->slot[0].maxiter || s.use_precond != t->slot[0].use_precond) {
            t->err = "smoqy_team_pff_step: the members of a round must pass the same tol / maxiter / use_precond";
            return 1;
        }
    }
    if (any_x)
        for (int w = 0; w < K; ++w)
            if (!t->x_seen[w]) { t->err = "smoqy_team_pff_step: a member passed x = NULL before it ever supplied its phonon fields"; return 1; }
    const int use_pre = t->slot[0].use_precond;
    if (use_pre && !all_rv) { t->err = "smoqy_team_pff_step: use_precond needs a Lanczos start vector from every member"; return 1; }
    if (int rc = smoqy_pff_step_v(c, t->phi, t->psi, any_x ? t->h_x : nullptr, use_pre ? t->h_rv : nullptr, t->slot[0].tol, t->slot[0].maxiter, use_pre, t->Sf.data(), t->iters.data(), t->eps.data(),
                                  any_force ? t->h_dS : nullptr)) {
        t->err = smoqy_last_error(c);
        return rc;
    }
    for (int w = 0; w < K; ++w) {
        const Slot &s = t->slot[w];
        if (s.Sf) *s.Sf = t->Sf[w];
        if (s.iters) *s.iters = t->iters[w];
        if (s.eps) *s.eps = t->eps[w];
    }
    return 0;
}

// deposit member w's arguments, wait for the others; the last arrival runs the round.  Every member copies its OWN arrays into / out of
// its part of the page-locked staging buffers outside the lock (K copies in parallel instead of K in a row inside the round): a member's
// part is its own between its return from one round and its arrival in the next, and no round starts before every member has arrived.
static int rendezvous(smoqy_team *t, int w, int op, const Slot &args)
{
    if (!t) return 1;
    if (w < 0 || w >= t->K || t->served) {
        std::lock_guard<std::mutex> lk(t->m);  // the error string is shared by the members
        t->err = t->served ? "this team is published (smoqy_team_serve): its members call smoqy_member_*" : "team member index out of range";
        return 1;
    }
    stage_in(t->stage(), t->K, t->Lt, t->N, t->Nph, w, args);
    const int rc = rendezvous_locked(t, w, op, args);
    if (rc == 0) stage_out(t->stage(), t->Lt, t->Nph, w, args);
    return rc;
}

static int rendezvous_locked(smoqy_team *t, int w, int op, const Slot &args)
{
    std::unique_lock<std::mutex> lk(t->m);
    if (t->arrived > 0 && t->op != op) { t->err = "team members made different calls in the same round"; return 8; }
    t->op = op;
    t->slot[w] = args;
    const unsigned long my_gen = t->gen;
    if (++t->arrived == t->K) {
        t->running = true;
        lk.unlock();
        const int rc = run_round(t);  // every other member is blocked below: the slots and the staging buffers are this thread's
        lk.lock();
        for (auto &s : t->slot) s.rc = rc;
        t->arrived = 0;
        t->op = OP_NONE;
        t->running = false;
        ++t->gen;
        lk.unlock();
        t->cv.notify_all();
        return rc;
    }
    // the deadline covers the wait for the OTHER MEMBERS only.  Once all K have arrived the round writes through every member's slot
    // (stack / caller temporaries) and staging part, so a member whose deadline passes while the round runs keeps waiting for the
    // round's result and is never taken out of `arrived` (leaving then was a use-after-return)
    const auto pred = [&] { return t->gen != my_gen; };
    if (!t->cv.wait_for(lk, std::chrono::duration<double>(t->timeout_s), pred)) {
        if (t->running) {
            t->cv.wait(lk, pred);
        } else {
            --t->arrived;  // give up this round (the caller rejects its update, as the reference's catch block does)
            t->err = "team rendezvous timed out: not every member made the call";
            return 9;
        }
    }
    return t->slot[w].rc;
}

extern "C" {

int smoqy_team_sample_phi(smoqy_team *t, int w, const void *R, double *RdotR)
{
    if (!R) return 1;
    Slot s;
    s.R = R; s.RdotR = RdotR;
    return rendezvous(t, w, OP_SAMPLE, s);
}

int smoqy_team_pff_step(smoqy_team *t, int w, const double *x, const double *randvec, double tol, int maxiter, int use_precond, double *Sf, int *iters, double *eps, double *dSdx)
{
    Slot s;
    s.x = x; s.rv = randvec; s.tol = tol; s.maxiter = maxiter; s.use_precond = use_precond ? 1 : 0;
    s.Sf = Sf; s.iters = iters; s.eps = eps; s.dSdx = dSdx;
    return rendezvous(t, w, OP_PFF, s);
}

int smoqy_team_hmc_update(smoqy_team *t, int w, const double *x, const void *R, const double *P, const double *randvecs, int Nt, double dt, double tol_force, double tol, int maxiter,
                          double *H0, double *H1, double *x_new, int *iters)
{
    if (!R || !P || !randvecs) return 1;
    if (Nt < 1 || Nt > kMaxNt) { if (t) t->err = "smoqy_team_hmc_update: Nt outside 1 … 64"; return 1; }
    Slot s;
    s.x = x; s.R = R; s.P = P; s.rvs = randvecs; s.Nt = Nt; s.dt = dt; s.tol_force = tol_force; s.tol = tol; s.maxiter = maxiter;
    s.H0 = H0; s.H1 = H1; s.x_new = x_new; s.iters = iters;
    return rendezvous(t, w, OP_HMC, s);
}

int smoqy_team_hmc_finish(smoqy_team *t, int w, int accept)
{
    Slot s;
    s.accept = accept ? 1 : 0;
    return rendezvous(t, w, OP_FINISH, s);
}

int smoqy_team_ge_config(smoqy_team *t, int Nrv, int n_orbitals, int D, const int64_t *Ldims)
{
    if (!t || Nrv < 1 || n_orbitals < 1 || D < 1 || !Ldims) return 1;
    std::lock_guard<std::mutex> lk(t->m);
    if (t->served) { t->err = "smoqy_team_ge_config: configure the GreensEstimator before smoqy_team_serve"; return 1; }
    if (t->ge) { t->err = "smoqy_team_ge_config: already configured"; return 1; }
    size_t Nc = 1;
    for (int d = 0; d < D; ++d) Nc *= (size_t)(Ldims[d] > 0 ? Ldims[d] : 0);
    smoqy_ctx *ge = nullptr;
    if (int rc = smoqy_clone(&ge, t->c, Nrv)) { t->err = std::string("smoqy_team_ge_config: ") + smoqy_last_error(nullptr); return rc; }
    int rc = smoqy_ge_config(ge, n_orbitals, D, Ldims);
    int v[3] = {-1, -1, -1};
    for (int q = 0; q < 3 && !rc; ++q) rc = smoqy_vec_alloc(ge, &v[q]);
    const size_t nR = (size_t)t->Lt * t->N * 16, gbytes = ((size_t)t->Lt + 1) * Nc * 16;
    void *gr = nullptr, *g = nullptr;
    if (!rc) rc = smoqy_host_alloc(t->c, &gr, nR * Nrv * t->K);
    if (!rc) rc = smoqy_host_alloc(t->c, &g, gbytes * t->K);
    if (rc) {
        t->err = std::string("smoqy_team_ge_config: ") + smoqy_last_error(ge);
        if (gr) smoqy_host_free(t->c, gr);
        if (g) smoqy_host_free(t->c, g);
        smoqy_destroy(ge);
        return rc;
    }
    t->ge = ge; t->ge_Nrv = Nrv; t->ge_r = v[0]; t->ge_gr = v[1]; t->ge_mtr = v[2]; t->ge_gbytes = gbytes;
    t->h_GR = (char *)gr; t->h_G = (char *)g;
    t->ge_it.resize((size_t)t->K * Nrv); t->ge_eps.resize((size_t)t->K * Nrv);
    return 0;
}

int smoqy_team_ge_update(smoqy_team *t, int w, const void *R, const double *randvec, double tol, int maxiter, int *iters, double *eps)
{
    if (!R || !randvec) return 1;
    Slot s;
    s.Rrv = R; s.rv = randvec; s.tol = tol; s.maxiter = maxiter; s.iters = iters; s.eps = eps;
    return rendezvous(t, w, OP_GE_UPDATE, s);
}

int smoqy_team_ge_measure_GD0(smoqy_team *t, int w, int a, int b, void *out)
{
    if (!out) return 1;
    Slot s;
    s.orb_a = a; s.orb_b = b; s.G = out;
    return rendezvous(t, w, OP_GE_GD0, s);
}


}  // extern "C"

// ---- teams across processes ---------------------------------------------------------------------------------------------------------
// The reference's walkers are MPI ranks: processes, not threads.  smoqy_team_serve publishes a team in a POSIX shared-memory segment;
// a rank of the same node joins with smoqy_member_attach (no GPU, no handle on its side) and makes the same two calls.  A member copies its
// arrays into its part of the segment's staging area (page-locked in the serving process, so the batched upload reads it directly),
// deposits its scalars and sleeps on a process-shared condition variable; a server thread in the process that owns the handle runs the
// round once all K members have arrived — the same run_round as the in-process team, its slots pointing into the segment.

struct Served {
    std::string name;
    ShmHeader *h = nullptr;
    std::thread server;
    void *own_R = nullptr;
    double *own_x = nullptr, *own_rv = nullptr, *own_dS = nullptr, *own_P = nullptr, *own_rvs = nullptr;  // the team's private staging, put back by unserve
    char *own_GR = nullptr, *own_G = nullptr;
    bool registered = false;
};


static void serve_loop(smoqy_team *t)
{
    ShmHeader *h = t->served->h;
    ShmMember *mem = (ShmMember *)((char *)h + h->off_members);
    shm_lock(h);
    for (;;) {
        while (!h->shutdown && h->arrived < h->K) pthread_cond_wait(&h->cv_arrive, &h->m);
        if (h->shutdown) break;
        t->op = h->op;
        for (int w = 0; w < t->K; ++w) {
            ShmMember &q = mem[w];
            Slot s;
            // run_round reads the array pointers as flags only (the data already sits in the staging area, which IS the segment)
            s.R = q.has_R ? (const void *)h : nullptr;
            s.x = q.has_x ? (const double *)h : nullptr;
            s.rv = q.has_rv ? (const double *)h : nullptr;
            s.P = q.has_P ? (const double *)h : nullptr;
            s.rvs = q.has_rvs ? (const double *)h : nullptr;
            s.dSdx = q.want_force ? (double *)h : nullptr;
            s.x_new = q.want_xnew ? (double *)h : nullptr;
            s.Rrv = q.has_Rrv ? (const void *)h : nullptr;
            s.G = q.want_G ? (void *)h : nullptr;
            s.orb_a = q.orb_a; s.orb_b = q.orb_b;
            s.tol = q.tol; s.maxiter = q.maxiter; s.use_precond = q.use_precond;
            s.Nt = q.Nt; s.dt = q.dt; s.tol_force = q.tol_force; s.accept = q.accept;
            s.Sf = &q.Sf; s.iters = &q.iters; s.eps = &q.eps; s.RdotR = &q.RdotR; s.H0 = q.H0; s.H1 = q.H1;
            t->slot[w] = s;
        }
        h->running = 1;  // from here to the broadcast below a member's deadline does not apply (member_round)
        pthread_mutex_unlock(&h->m);
        const int rc = run_round(t);  // every member is asleep on cv_done
        shm_lock(h);
        h->running = 0;
        for (int w = 0; w < t->K; ++w) mem[w].rc = rc;
        h->rc = rc;
        snprintf(h->err, sizeof(h->err), "%s", rc ? t->err.c_str() : "");
        h->arrived = 0;
        h->op = OP_NONE;
        t->op = OP_NONE;
        ++h->gen;
        pthread_cond_broadcast(&h->cv_done);
    }
    pthread_mutex_unlock(&h->m);
}

extern "C" {

int smoqy_team_serve(smoqy_team *t, const char *name, const double *x0)
{
    if (!t || !name || name[0] != '/') { g_team_error = "smoqy_team_serve: null team or a name that does not start with '/'"; return 1; }
    if (t->served) { t->err = "smoqy_team_serve: this team is already published"; return 1; }
    const size_t nR = (size_t)t->Lt * t->N * 16, nx = (size_t)std::max(t->Nph, 1) * t->Lt * sizeof(double);
    ShmHeader lay{};
    lay.off_members = align_up(sizeof(ShmHeader), 64);
    lay.off_R = align_up(lay.off_members + sizeof(ShmMember) * (size_t)t->K, 4096);
    lay.off_x = align_up(lay.off_R + nR * t->K, 4096);
    lay.off_rv = align_up(lay.off_x + nx * t->K, 4096);
    lay.off_dS = align_up(lay.off_rv + (size_t)t->N * t->K * sizeof(double), 4096);
    lay.off_P = align_up(lay.off_dS + nx * t->K, 4096);
    lay.off_rvs = align_up(lay.off_P + nx * t->K, 4096);
    lay.off_GR = align_up(lay.off_rvs + (size_t)t->N * t->K * (kMaxNt + 1) * sizeof(double), 4096);
    lay.off_G = align_up(lay.off_GR + nR * t->ge_Nrv * t->K, 4096);  // both empty without smoqy_team_ge_config
    lay.total = align_up(lay.off_G + t->ge_gbytes * t->K, 4096);
    shm_unlink(name);  // a stale segment of a crashed job
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { t->err = std::string("smoqy_team_serve: shm_open failed for ") + name; return 2; }
    if (ftruncate(fd, (off_t)lay.total) != 0) { close(fd); shm_unlink(name); t->err = "smoqy_team_serve: ftruncate failed (is /dev/shm large enough?)"; return 2; }
    void *p = mmap(nullptr, lay.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); t->err = "smoqy_team_serve: mmap failed"; return 2; }
    ShmHeader *h = (ShmHeader *)p;
    std::memset(h, 0, lay.off_R);
    h->K = t->K; h->Lt = t->Lt; h->N = t->N; h->Nph = t->Nph;
    h->timeout_s = t->timeout_s;
    h->server_pid = (int)getpid();
    h->off_members = lay.off_members; h->off_R = lay.off_R; h->off_x = lay.off_x; h->off_rv = lay.off_rv; h->off_dS = lay.off_dS; h->off_P = lay.off_P; h->off_rvs = lay.off_rvs; h->off_GR = lay.off_GR; h->off_G = lay.off_G; h->total = lay.total;
    h->ge_Nrv = t->ge_Nrv; h->ge_gbytes = t->ge_gbytes;
    pthread_mutexattr_t ma;
    pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
    pthread_mutex_init(&h->m, &ma);
    pthread_mutexattr_destroy(&ma);
    pthread_condattr_t ca;
    pthread_condattr_init(&ca);
    pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
    pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
    pthread_cond_init(&h->cv_arrive, &ca);
    pthread_cond_init(&h->cv_done, &ca);
    pthread_condattr_destroy(&ca);
    Served *sv = new Served();
    sv->name = name; sv->h = h;
    sv->own_R = t->h_R; sv->own_x = t->h_x; sv->own_rv = t->h_rv; sv->own_dS = t->h_dS; sv->own_P = t->h_P; sv->own_rvs = t->h_rvs; sv->own_GR = t->h_GR; sv->own_G = t->h_G;
    char *base = (char *)p;
    // the staging area of the team now IS the segment; page-lock it where the driver allows (transfers work either way)
    sv->registered = smoqy_host_register(t->c, base + h->off_R, h->total - h->off_R) == 0;
    t->h_R = base + h->off_R; t->h_x = (double *)(base + h->off_x); t->h_rv = (double *)(base + h->off_rv); t->h_dS = (double *)(base + h->off_dS);
    t->h_P = (double *)(base + h->off_P); t->h_rvs = (double *)(base + h->off_rvs);
    if (t->ge) { t->h_GR = base + h->off_GR; t->h_G = base + h->off_G; }
    std::memcpy(t->h_x, x0 ? (const void *)x0 : (const void *)sv->own_x, nx * t->K);
    t->served = sv;
    sv->server = std::thread(serve_loop, t);
    __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE);  // members wait for this
    return 0;
}

int smoqy_team_unserve(smoqy_team *t)
{
    if (!t || !t->served) return 0;
    Served *sv = t->served;
    ShmHeader *h = sv->h;
    shm_lock(h);
    h->shutdown = 1;
    snprintf(h->err, sizeof(h->err), "the team was withdrawn by its serving process");
    for (int w = 0; w < h->K; ++w) ((ShmMember *)((char *)h + h->off_members))[w].rc = 10;
    ++h->gen;
    pthread_cond_broadcast(&h->cv_arrive);
    pthread_cond_broadcast(&h->cv_done);
    pthread_mutex_unlock(&h->m);
    if (sv->server.joinable()) sv->server.join();
    if (sv->registered) smoqy_host_unregister(t->c, (char *)h + h->off_R);
    t->h_R = sv->own_R; t->h_x = sv->own_x; t->h_rv = sv->own_rv; t->h_dS = sv->own_dS; t->h_P = sv->own_P; t->h_rvs = sv->own_rvs; t->h_GR = sv->own_GR; t->h_G = sv->own_G;
    shm_unlink(sv->name.c_str());
    munmap(h, h->total);
    delete sv;
    t->served = nullptr;
    return 0;
}

}  // extern "C"

extern "C" {

// ---- measurement: K native member threads ----------------------------------------------------------------------------------------
// What a caller without an interpreter lock gets from a team: K std::threads, each running the per-walker sweep of the reference's
// tutorial (tutorials/holstein_honeycomb.jl:611-684: two local-move-like updates, then an HMC trajectory of Nt force evaluations and the
// closing action) against ITS OWN walker through smoqy_team_sample_phi / smoqy_team_pff_step, with its own random stream (xoshiro256++
// and a ziggurat: the randn! calls of src/PFFCalculator.jl:67 and src/KPMPreconditioner.jl:634 are part of a member's host work).  The
// phonon-field moves are the synthetic drift of bench.py's sweep (x += drift·π, restored afterwards), as in walkers.TeamMember.sweep.
namespace {
struct Xo {
    uint64_t s[4];
    explicit Xo(uint64_t seed) { for (auto &v : s) { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = z ^ (z >> 31); } }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() { const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    // standard normal deviates by the ziggurat method (128 layers; Marsaglia & Tsang 2000 with Doornik's separate draws for the layer
    // index and the abscissa): a few ns per number, the class of generator behind randn! in the reference's driver
    struct Zig {
        double x[129], ratio[128];
        Zig()
        {
            const double R = 3.442619855899, V = 9.91256303526217e-3;
            double f = std::exp(-0.5 * R * R);
            x[0] = V / f; x[1] = R; x[128] = 0.0;
            for (int i = 2; i < 128; ++i) { x[i] = std::sqrt(-2.0 * std::log(V / x[i - 1] + f)); f = std::exp(-0.5 * x[i] * x[i]); }
            for (int i = 0; i < 128; ++i) ratio[i] = x[i + 1] / x[i];
        }
    };
    double one()
    {
        static const Zig z;
        for (;;) {
            const double u = 2.0 * uni() - 1.0;
            const int i = (int)(next() >> 57);  // 7 bits of their own
            if (std::fabs(u) < z.ratio[i]) return u * z.x[i];
            if (i == 0) {  // the tail beyond R
                double a, b;
                do { a = std::log(uni()) / z.x[1]; b = std::log(uni()); } while (-2.0 * b < a * a);
                return u < 0 ? a - z.x[1] : z.x[1] - a;
            }
            const double v = u * z.x[i];
            const double f0 = std::exp(-0.5 * (z.x[i] * z.x[i] - v * v)), f1 = std::exp(-0.5 * (z.x[i + 1] * z.x[i + 1] - v * v));
            if (f1 + uni() * (f0 - f1) < 1.0) return v;
        }
    }
    void normal(double *out, size_t n, double scale)
    {
        for (size_t i = 0; i < n; ++i) out[i] = scale * one();
    }
};
// what one sweep of a member asks its generator for, in the order it asks: drawn by the member's producer thread one sweep ahead, so that
// a member's randn! calls run while the device works on the team's current call instead of between two calls (the lock-step batches of
// bench.py do the same: WalkerBatch(prefetch_randoms)); the stream of numbers is the one the member would draw on demand
struct Draws {
    std::vector<double> R[3], pi[3], P, rvs;
    std::vector<std::vector<double>> rv;
};
struct Member {
    std::vector<double> x, dS;
    Xo rng{0};
    long solves = 0, iters = 0;
    int rc = 0;
    // two draw sets: the producer fills set (n & 1) for sweep n as soon as sweep n - 2 has released it
    Draws d[2];
    std::mutex mu;
    std::condition_variable cv;
    long produced = 0, consumed = 0;
    bool stop = false;
};
}  // namespace

int smoqy_bench_randn(double *out, long n, unsigned long seed, double scale)
{
    if (!out || n < 0) return 1;
    Xo rng((uint64_t)seed);
    rng.normal(out, (size_t)n, scale);
    return 0;
}

int smoqy_team_bench_sweeps(smoqy_team *t, const double *x0, int nfree, double drift, int Nt, double tol, double tol_force, int maxiter, int device_hmc, int warmup_sweeps, int nsweeps,
                            unsigned long seed, double *seconds, long *solves, long *iters)
{
    if (!t || !x0 || nfree < 0 || nfree > t->Nph || Nt < 1 || nsweeps < 1 || !seconds || (device_hmc && Nt > kMaxNt)) return 1;
    const int K = t->K, Lt = t->Lt, N = t->N, Nph = t->Nph;
    const size_t nx = (size_t)Nph * Lt;
    const int nrv = device_hmc ? 2 : 2 + Nt + 1, npi = device_hmc ? 2 : 3;  // start vectors / momenta a sweep draws one at a time
    std::vector<Member> mem((size_t)K);
    for (int w = 0; w < K; ++w) {
        Member &m = mem[w];
        m.x.assign(x0 + (size_t)w * nx, x0 + (size_t)(w + 1) * nx);
        m.dS.resize(nx);
        for (Draws &d : m.d) {
            for (auto &R : d.R) R.resize(2 * (size_t)Lt * N);
            for (int q = 0; q < npi; ++q) d.pi[q].resize((size_t)Lt * nfree);
            d.rv.assign((size_t)nrv, std::vector<double>((size_t)N));
            if (device_hmc) { d.P.resize(nx); d.rvs.resize((size_t)N * (Nt + 1)); }
        }
        m.rng = Xo(seed + 7919ull * (uint64_t)w);
    }
    // the member's random stream, in the order of the sweep below
    auto draw = [&](Member &m, Draws &d) {
        const double h = std::sqrt(0.5);
        for (int rep = 0; rep < 2; ++rep) {
            m.rng.normal(d.R[rep].data(), d.R[rep].size(), h);        // randn!(rng, Φ), src/PFFCalculator.jl:67
            m.rng.normal(d.pi[rep].data(), d.pi[rep].size(), 1.0);
            m.rng.normal(d.rv[rep].data(), d.rv[rep].size(), 1.0);    // randn!(rng, v), src/KPMPreconditioner.jl:634
        }
        m.rng.normal(d.R[2].data(), d.R[2].size(), h);
        if (device_hmc) {
            m.rng.normal(d.P.data(), d.P.size(), 1.0);
            m.rng.normal(d.rvs.data(), d.rvs.size(), 1.0);
        } else {
            m.rng.normal(d.pi[2].data(), d.pi[2].size(), 1.0);
            for (int q = 2; q < nrv; ++q) m.rng.normal(d.rv[q].data(), d.rv[q].size(), 1.0);
        }
    };
    auto shift = [&](Member &m, const std::vector<double> &pi, double f) {  // x[:, :nfree] += f·π
        for (int l = 0; l < Lt; ++l)
            for (int j = 0; j < nfree; ++j) m.x[(size_t)l * Nph + j] += f * pi[(size_t)l * nfree + j];
    };
    auto step = [&](Member &m, int w, const std::vector<double> &rv, double tl, bool force) {
        double sf = 0, eps = 0;
        int it = 0;
        const int rc = smoqy_team_pff_step(t, w, m.x.data(), rv.data(), tl, maxiter, 1, &sf, &it, &eps, force ? m.dS.data() : nullptr);
        m.solves += 1; m.iters += it;
        return rc;
    };
    auto sample = [&](int w, const std::vector<double> &R) {
        double rr = 0;
        return smoqy_team_sample_phi(t, w, R.data(), &rr);
    };
    auto sweep = [&](Member &m, int w, Draws &d) {
        for (int rep = 0; rep < 2; ++rep) {
            if (int rc = sample(w, d.R[rep])) return rc;
            shift(m, d.pi[rep], drift);
            if (int rc = step(m, w, d.rv[rep], tol, false)) return rc;
            shift(m, d.pi[rep], -drift);
        }
        if (device_hmc) {
            // hmc_update! with the trajectory on the device (smoqy_team_hmc_update), Δt = π/(2 Nt) (tutorials/holstein_honeycomb.jl:542);
            // always rejected, as bench.py's sweep does
            double H0[3], H1[3];
            int it = 0;
            if (int rc = smoqy_team_hmc_update(t, w, m.x.data(), d.R[2].data(), d.P.data(), d.rvs.data(), Nt, 1.5707963267948966 / Nt, tol_force, tol, maxiter, H0, H1, nullptr, &it)) return rc;
            m.solves += Nt + 1; m.iters += it;
            return smoqy_team_hmc_finish(t, w, 0);
        }
        if (int rc = sample(w, d.R[2])) return rc;
        for (int q = 0; q < Nt; ++q) {
            if (int rc = step(m, w, d.rv[2 + q], tol_force, true)) return rc;
            shift(m, d.pi[2], drift / Nt);
        }
        if (int rc = step(m, w, d.rv[2 + Nt], tol, false)) return rc;
        shift(m, d.pi[2], -drift);
        return 0;
    };
    // producers: one per member, alive over warm-up and timed sweeps, at most two sweeps ahead of their member
    std::vector<std::thread> producers;
    for (int w = 0; w < K; ++w)
        producers.emplace_back([&, w] {
            Member &m = mem[w];
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(m.mu);
                    m.cv.wait(lk, [&] { return m.stop || m.produced - m.consumed < 2; });
                    if (m.stop) return;
                }
                draw(m, m.d[m.produced & 1]);
                {
                    std::lock_guard<std::mutex> lk(m.mu);
                    ++m.produced;
                }
                m.cv.notify_all();
            }
        });
    auto run = [&](int count) {
        std::vector<std::thread> th;
        for (int w = 0; w < K; ++w)
            th.emplace_back([&, w] {
                Member &m = mem[w];
                for (int q = 0; q < count && m.rc == 0; ++q) {
                    {
                        std::unique_lock<std::mutex> lk(m.mu);
                        m.cv.wait(lk, [&] { return m.produced > m.consumed; });
                    }
                    m.rc = sweep(m, w, m.d[m.consumed & 1]);
                    {
                        std::lock_guard<std::mutex> lk(m.mu);
                        ++m.consumed;
                    }
                    m.cv.notify_all();
                }
            });
        for (auto &x : th) x.join();
    };
    struct StopProducers {  // on every way out, errors included
        std::vector<Member> &mem; std::vector<std::thread> &th;
        ~StopProducers()
        {
            for (auto &m : mem) { { std::lock_guard<std::mutex> lk(m.mu); m.stop = true; } m.cv.notify_all(); }
            for (auto &x : th) x.join();
        }
    } stop_producers{mem, producers};
    if (warmup_sweeps > 0) run(warmup_sweeps);
    for (auto &m : mem) { m.solves = 0; m.iters = 0; }
    const auto t0 = std::chrono::steady_clock::now();
    run(nsweeps);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long so = 0, itn = 0;
    int rc = 0;
    for (auto &m : mem) { so += m.solves; itn += m.iters; if (m.rc) rc = m.rc; }
    if (solves) *solves = so;
    if (iters) *iters = itn;
    return rc;
}

}  // extern "C"
