// Shared between the serving side of a walker team (team.hip, inside libsmoqy_hip.so) and the member side (member.cpp, which is also
// built alone as libsmoqy_member.so — no HIP, no rocFFT — for ranks that never touch a GPU): the per-call argument slot, the staging
// layout and the layout of the POSIX shared-memory segment a published team lives in.  Plain C++; nothing here calls the HIP runtime.
#pragma once
#include <cstdint>
#include <cstring>

#include <errno.h>
#include <pthread.h>
#include <time.h>

namespace smoqy_team_detail {
enum { OP_NONE = 0, OP_SAMPLE = 1, OP_PFF = 2, OP_HMC = 3, OP_FINISH = 4, OP_GE_UPDATE = 5, OP_GE_GD0 = 6 };
constexpr int kMaxNt = 64;  // leapfrog steps a team's staging is sized for (smoqy_team_hmc_update)
struct Slot {
    const void *R = nullptr;
    const double *x = nullptr, *rv = nullptr;
    double tol = 0;
    int maxiter = 0, use_precond = 0;
    double *Sf = nullptr, *eps = nullptr, *dSdx = nullptr, *RdotR = nullptr;
    int *iters = nullptr;
    // OP_HMC: momentum deviates, N x (Nt + 1) Lanczos start vectors, trajectory parameters, {S_f, S_b, K} before / after, proposed fields
    const double *P = nullptr, *rvs = nullptr;
    int Nt = 0;
    double dt = 0, tol_force = 0;
    double *H0 = nullptr, *H1 = nullptr, *x_new = nullptr;
    int accept = 0;  // OP_FINISH
    // GreensEstimator: Nrv random vectors (Ltau x N x Nrv, OP_GE_UPDATE); orbitals and the member's G(Δ,0) array (OP_GE_GD0)
    const void *Rrv = nullptr;
    void *G = nullptr;
    int orb_a = 0, orb_b = 0;
    int rc = 0;
};
// where a member's arrays are staged for the batched call: the team's own page-locked buffers, or the shared-memory segment
struct Stage {
    char *R = nullptr;
    double *x = nullptr, *rv = nullptr, *dS = nullptr, *P = nullptr, *rvs = nullptr;
    char *GR = nullptr, *G = nullptr;  // GreensEstimator: the members' random vectors (Nrv per member) and their G(Δ,0) arrays
    int Nrv = 0;
    size_t gbytes = 0;                 // bytes of one member's G(Δ,0)
};
// a member copies its OWN inputs in before the rendezvous and its own outputs out after it (K copies in parallel, outside any lock)
inline void stage_in(const Stage &g, int K, int Lt, int N, int Nph, int w, const Slot &a)
{
    const size_t nR = (size_t)Lt * N * 16, nx = (size_t)(Nph > 0 ? Nph : 1) * Lt;
    if (a.R) std::memcpy(g.R + (size_t)w * nR, a.R, nR);
    if (a.x) std::memcpy(g.x + (size_t)w * nx, a.x, nx * sizeof(double));
    if (a.rv) std::memcpy(g.rv + (size_t)w * N, a.rv, (size_t)N * sizeof(double));
    if (a.P) std::memcpy(g.P + (size_t)w * nx, a.P, nx * sizeof(double));
    if (a.Rrv && g.GR) std::memcpy(g.GR + (size_t)w * nR * g.Nrv, a.Rrv, nR * g.Nrv);
    if (a.rvs && a.Nt >= 1 && a.Nt <= kMaxNt)  // the batched trajectory wants N x K x Nt: step-major, member w's vector of step t at (t K + w) N
        for (int t = 0; t <= a.Nt; ++t) std::memcpy(g.rvs + ((size_t)t * K + w) * N, a.rvs + (size_t)t * N, (size_t)N * sizeof(double));
}
inline void stage_out(const Stage &g, int Lt, int Nph, int w, const Slot &a)
{
    const size_t nx = (size_t)(Nph > 0 ? Nph : 1) * Lt;
    if (a.dSdx) std::memcpy(a.dSdx, g.dS + (size_t)w * nx, nx * sizeof(double));
    if (a.x_new) std::memcpy(a.x_new, g.dS + (size_t)w * nx, nx * sizeof(double));  // OP_HMC returns the proposed fields through the force staging
    if (a.G && g.G) std::memcpy(a.G, g.G + (size_t)w * g.gbytes, g.gbytes);
}
constexpr uint64_t kShmMagic = 0x534d4f5159544d34ull;  // "SMOQYTM4" (layout version: bump with every change of ShmHeader / ShmMember)
struct ShmMember {
    int has_x, has_rv, want_force, maxiter, use_precond, iters, rc, attached;
    int has_R, has_P, has_rvs, want_xnew, Nt, accept;
    int has_Rrv, want_G, orb_a, orb_b;
    double tol, Sf, eps, RdotR;
    double dt, tol_force, H0[3], H1[3];
};
struct ShmHeader {
    uint64_t magic;
    int K, Lt, N, Nph;
    int op, arrived, shutdown, rc;
    int server_pid, pad0;  // the serving process (a member that waits on a running round checks that it is still alive)
    unsigned long gen;
    double timeout_s;
    size_t off_members, off_R, off_x, off_rv, off_dS, off_P, off_rvs, off_GR, off_G, total;
    int ge_Nrv;
    int running;      // the server thread is inside run_round: results go through every member's slot, nobody may leave the round
    size_t ge_gbytes;
    pthread_mutex_t m;
    pthread_cond_t cv_arrive, cv_done;
    char err[256];
};
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int shm_lock(ShmHeader *h)
{
    const int e = pthread_mutex_lock(&h->m);
    if (e == EOWNERDEAD) { pthread_mutex_consistent(&h->m); return 0; }  // a member died inside the lock: the state it guards is plain counters
    return e;
}
inline timespec deadline_after(double seconds)
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = ts.tv_sec + ts.tv_nsec * 1e-9 + seconds;
    ts.tv_sec = (time_t)t;
    ts.tv_nsec = (long)((t - (double)ts.tv_sec) * 1e9);
    return ts;
}
}  // namespace smoqy_team_detail
