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
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/smoqy_hip.h"

namespace {
enum { OP_NONE = 0, OP_SAMPLE = 1, OP_PFF = 2 };
struct Slot {
    const void *R = nullptr;
    const double *x = nullptr, *rv = nullptr;
    double tol = 0;
    int maxiter = 0, use_precond = 0;
    double *Sf = nullptr, *eps = nullptr, *dSdx = nullptr, *RdotR = nullptr;
    int *iters = nullptr;
    int rc = 0;
};
}  // namespace

struct smoqy_team {
    smoqy_ctx *c = nullptr;
    int K = 0, Lt = 0, N = 0, Nph = 0;
    int phi = -1, psi = -1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, op = OP_NONE;
    unsigned long gen = 0;
    double timeout_s = 600.0;
    std::vector<Slot> slot;
    std::vector<char> x_seen;  // member w has supplied its phonon fields at least once (later NULLs mean "unchanged")
    // page-locked staging in the batched layouts of the C ABI
    void *h_R = nullptr;                                     // Ltau x N x K complex128
    double *h_x = nullptr, *h_rv = nullptr, *h_dS = nullptr;  // Nph x Ltau x K, N x K, Nph x Ltau x K
    std::vector<double> Sf, eps, dot;
    std::vector<int> iters;
    std::string err;
};

static std::string g_team_error;

extern "C" {

const char *smoqy_team_last_error(const smoqy_team *t) { return t ? t->err.c_str() : g_team_error.c_str(); }

int smoqy_team_destroy(smoqy_team *t)
{
    if (!t) return 0;
    if (t->c) {
        if (t->h_R) smoqy_host_free(t->c, t->h_R);
        if (t->h_x) smoqy_host_free(t->c, t->h_x);
        if (t->h_rv) smoqy_host_free(t->c, t->h_rv);
        if (t->h_dS) smoqy_host_free(t->c, t->h_dS);
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
    smoqy_team *t = new smoqy_team();
    t->c = ctx; t->Lt = d[0]; t->N = d[1]; t->K = d[4]; t->Nph = Nph;
    t->slot.resize((size_t)t->K);
    t->x_seen.assign((size_t)t->K, 0);
    t->Sf.resize((size_t)t->K); t->eps.resize((size_t)t->K); t->iters.resize((size_t)t->K); t->dot.resize(2 * (size_t)t->K);
    const size_t nR = (size_t)t->Lt * t->N * t->K * 16, nx = (size_t)std::max(Nph, 1) * t->Lt * t->K * sizeof(double);
    int rc = smoqy_host_alloc(ctx, &t->h_R, nR);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_x, nx);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_dS, nx);
    if (!rc) rc = smoqy_host_alloc(ctx, (void **)&t->h_rv, (size_t)t->N * t->K * sizeof(double));
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

// the batched call of one round, run by the last member to arrive (all the others are blocked in rendezvous)
static int run_round(smoqy_team *t)
{
    smoqy_ctx *c = t->c;
    const int K = t->K;
    const size_t nR = (size_t)t->Lt * t->N * 16, nx = (size_t)t->Nph * t->Lt;
    if (t->op == OP_SAMPLE) {
        // sample_pseudofermion_fields! (src/PFFCalculator.jl:56-76): Φ = Λᵀ Mᵀ R, |R|² per member
        for (int w = 0; w < K; ++w) std::memcpy((char *)t->h_R + (size_t)w * nR, t->slot[w].R, nR);
        if (int rc = smoqy_vec_upload(c, t->phi, t->h_R, 0, K)) return rc;
        if (int rc = smoqy_vec_dot(c, t->phi, t->phi, t->dot.data())) return rc;
        if (int rc = smoqy_matvec_v(c, SMOQY_OP_MT, t->phi, t->phi)) return rc;                // lmul_Mt! (:71)
        if (int rc = smoqy_lambda_apply_v(c, SMOQY_LAMBDA_MULT, t->phi, t->phi)) return rc;    // mul_Λᵀ! (:73)
        for (int w = 0; w < K; ++w)
            if (t->slot[w].RdotR) *t->slot[w].RdotR = t->dot[2 * (size_t)w];
        return 0;
    }
    // calculate_fermionic_action! / calculate_derivative_fermionic_action! (src/PFFCalculator.jl:79-157) behind the field update of
    // the caller's move (src/EFAPFFHMCUpdater.jl:200-205, src/reflection_update.jl:99)
    bool any_x = false, any_force = false, all_rv = true;
    for (int w = 0; w < K; ++w) {
        const Slot &s = t->slot[w];
        if (s.x) { std::memcpy(t->h_x + (size_t)w * nx, s.x, nx * sizeof(double)); t->x_seen[w] = 1; any_x = true; }
        if (s.dSdx) any_force = true;
        if (s.rv) std::memcpy(t->h_rv + (size_t)w * t->N, s.rv, (size_t)t->N * sizeof(double));
        else all_rv = false;
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
        if (s.dSdx) std::memcpy(s.dSdx, t->h_dS + (size_t)w * nx, nx * sizeof(double));
    }
    return 0;
}

// deposit member w's arguments, wait for the others; the last arrival runs the round
static int rendezvous(smoqy_team *t, int w, int op, const Slot &args)
{
    if (!t) return 1;
    if (w < 0 || w >= t->K) { t->err = "team member index out of range"; return 1; }
    std::unique_lock<std::mutex> lk(t->m);
    if (t->arrived > 0 && t->op != op) { t->err = "team members made different calls in the same round"; return 8; }
    t->op = op;
    t->slot[w] = args;
    const unsigned long my_gen = t->gen;
    if (++t->arrived == t->K) {
        lk.unlock();
        const int rc = run_round(t);  // every other member is blocked below: the slots and the staging buffers are this thread's
        lk.lock();
        for (auto &s : t->slot) s.rc = rc;
        t->arrived = 0;
        t->op = OP_NONE;
        ++t->gen;
        lk.unlock();
        t->cv.notify_all();
        return rc;
    }
    if (!t->cv.wait_for(lk, std::chrono::duration<double>(t->timeout_s), [&] { return t->gen != my_gen; })) {
        --t->arrived;  // give up this round (the caller rejects its update, as the reference's catch block does)
        t->err = "team rendezvous timed out: not every member made the call";
        return 9;
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

}  // extern "C"
