// TEST INFRASTRUCTURE, not product: a CPU stand-in for the serving side of a published walker team (smoqy_team_serve in
// smoqyelphqmc.jl_amd/csrc/team.hip), so that the MEMBER side — member.cpp / libsmoqy_member.so, the library a GPU-less rank of the
// reference's one-walker-per-rank model links (tutorials/holstein_honeycomb_mpi.jl:60-72) — can be exercised on the CPU by several
// processes: attach / index ownership, staging layout, the rendezvous, its deadline, a round that outlasts the deadline, a server that
// dies inside a round, a withdrawn team, a failing round.  It shares team_shm.h (the segment layout and the protocol's fields) with the
// product and NOTHING else; the "results" of a round are simple sums of what the members staged, so a member can check that its own
// data — and nobody else's — went through its part of the segment.
//
//   fake_team_server <name> <K> <Lt> <N> <Nph> <timeout_s> <mode>
//     mode  serve   serve rounds until SIGTERM (then withdraw the team like smoqy_team_unserve) or until stdin closes
//           slow    as serve, every round takes 3 x timeout_s (a member's deadline must not apply to a running round)
//           die     take the first round (running = 1), then _exit(3) without answering
// Results of a round, member w (nR = Lt N complex, nx = max(Nph,1) Lt doubles):
//   OP_SAMPLE  RdotR = sum |R_w|^2
//   OP_PFF     Sf = sum x_w^2 + sum rv_w, iters = 10 + w, eps = tol / 2, dS = 2 x_w; tol < 0 fails the round (rc 7, "fake failure")
//   OP_HMC     H0 = {sum P_w^2, sum x_w^2, Nt}, H1 = {sum over t <= Nt of rvs_w[t], dt, tol_force}, x_new = x_w + dt P_w, iters = Nt
//   OP_FINISH  accept: the proposed fields (still in the force staging) become the member's fields
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <string>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../smoqyelphqmc.jl_amd/csrc/team_shm.h"

using namespace smoqy_team_detail;

static volatile sig_atomic_t g_stop = 0;
static void on_term(int) { g_stop = 1; }

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s name K Lt N Nph timeout mode\n", argv[0]); return 2; }
    const std::string name = argv[1], mode = argv[7];
    const int K = atoi(argv[2]), Lt = atoi(argv[3]), N = atoi(argv[4]), Nph = atoi(argv[5]);
    const double timeout_s = atof(argv[6]);
    const size_t nR = (size_t)Lt * N * 16, nxd = (size_t)(Nph > 0 ? Nph : 1) * Lt, nx = nxd * sizeof(double);
    ShmHeader lay{};
    lay.off_members = align_up(sizeof(ShmHeader), 64);
    lay.off_R = align_up(lay.off_members + sizeof(ShmMember) * (size_t)K, 4096);
    lay.off_x = align_up(lay.off_R + nR * K, 4096);
    lay.off_rv = align_up(lay.off_x + nx * K, 4096);
    lay.off_dS = align_up(lay.off_rv + (size_t)N * K * sizeof(double), 4096);
    lay.off_P = align_up(lay.off_dS + nx * K, 4096);
    lay.off_rvs = align_up(lay.off_P + nx * K, 4096);
    lay.off_GR = align_up(lay.off_rvs + (size_t)N * K * (kMaxNt + 1) * sizeof(double), 4096);
    lay.off_G = lay.off_GR;
    lay.total = align_up(lay.off_G, 4096);
    shm_unlink(name.c_str());
    const int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)lay.total) != 0) { perror("shm"); return 2; }
    void *p = mmap(nullptr, lay.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { perror("mmap"); return 2; }
    ShmHeader *h = (ShmHeader *)p;
    std::memset(h, 0, lay.off_R);
    h->K = K; h->Lt = Lt; h->N = N; h->Nph = Nph;
    h->timeout_s = timeout_s;
    h->server_pid = (int)getpid();
    h->off_members = lay.off_members; h->off_R = lay.off_R; h->off_x = lay.off_x; h->off_rv = lay.off_rv; h->off_dS = lay.off_dS;
    h->off_P = lay.off_P; h->off_rvs = lay.off_rvs; h->off_GR = lay.off_GR; h->off_G = lay.off_G; h->total = lay.total;
    pthread_mutexattr_t ma;
    pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
    pthread_mutex_init(&h->m, &ma);
    pthread_condattr_t ca;
    pthread_condattr_init(&ca);
    pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
    pthread_condattr_setclock(&ca, CLOCK_MONOTONIC);
    pthread_cond_init(&h->cv_arrive, &ca);
    pthread_cond_init(&h->cv_done, &ca);
    char *base = (char *)p;
    double *X = (double *)(base + h->off_x), *RV = (double *)(base + h->off_rv), *DS = (double *)(base + h->off_dS), *P = (double *)(base + h->off_P),
           *RVS = (double *)(base + h->off_rvs);
    for (int w = 0; w < K; ++w)
        for (size_t i = 0; i < nxd; ++i) X[(size_t)w * nxd + i] = 100.0 * w + (double)i;  // the "initial phonon fields" smoqy_member_fields hands out
    ShmMember *mem = (ShmMember *)(base + h->off_members);
    struct sigaction sa{};
    sa.sa_handler = on_term;
    sigaction(SIGTERM, &sa, nullptr);
    __atomic_store_n(&h->magic, kShmMagic, __ATOMIC_RELEASE);
    printf("READY\n");
    fflush(stdout);
    shm_lock(h);
    while (!g_stop) {
        // (a timed wait so that SIGTERM is noticed; the product's server thread is woken by smoqy_team_unserve instead)
        while (!g_stop && h->arrived < h->K) {
            timespec dl = deadline_after(0.05);
            const int e = pthread_cond_timedwait(&h->cv_arrive, &h->m, &dl);
            if (e == EOWNERDEAD) pthread_mutex_consistent(&h->m);
        }
        if (g_stop) break;
        const int op = h->op;
        h->running = 1;
        pthread_mutex_unlock(&h->m);
        if (mode == "die") _exit(3);
        if (mode == "slow") usleep((useconds_t)(3.0 * timeout_s * 1e6));
        int rc = 0;
        std::string err;
        for (int w = 0; w < K && !rc; ++w) {
            ShmMember &q = mem[w];
            const double *x = X + (size_t)w * nxd;
            if (op == OP_SAMPLE) {
                const double *R = (const double *)(base + h->off_R + (size_t)w * nR);
                double s = 0;
                for (size_t i = 0; i < 2 * (size_t)Lt * N; ++i) s += R[i] * R[i];
                q.RdotR = s;
            } else if (op == OP_PFF) {
                if (q.tol < 0) { rc = 7; err = "fake failure"; break; }
                double s = 0;
                for (size_t i = 0; i < nxd; ++i) s += x[i] * x[i];
                if (q.has_rv) for (int i = 0; i < N; ++i) s += RV[(size_t)w * N + i];
                q.Sf = s; q.iters = 10 + w; q.eps = 0.5 * q.tol;
                if (q.want_force) for (size_t i = 0; i < nxd; ++i) DS[(size_t)w * nxd + i] = 2.0 * x[i];
            } else if (op == OP_HMC) {
                double sp = 0, sx = 0, sr = 0;
                for (size_t i = 0; i < nxd; ++i) { sp += P[(size_t)w * nxd + i] * P[(size_t)w * nxd + i]; sx += x[i] * x[i]; }
                for (int t = 0; t <= q.Nt; ++t) for (int i = 0; i < N; ++i) sr += RVS[((size_t)t * K + w) * N + i];
                q.H0[0] = sp; q.H0[1] = sx; q.H0[2] = q.Nt;
                q.H1[0] = sr; q.H1[1] = q.dt; q.H1[2] = q.tol_force;
                q.iters = q.Nt;
                for (size_t i = 0; i < nxd; ++i) DS[(size_t)w * nxd + i] = x[i] + q.dt * P[(size_t)w * nxd + i];
            } else if (op == OP_FINISH) {
                if (q.accept) std::memcpy(X + (size_t)w * nxd, DS + (size_t)w * nxd, nx);
            } else {
                rc = 1; err = "fake server: operation not implemented";
            }
        }
        shm_lock(h);
        h->running = 0;
        for (int w = 0; w < K; ++w) mem[w].rc = rc;
        h->rc = rc;
        snprintf(h->err, sizeof(h->err), "%s", err.c_str());
        h->arrived = 0;
        h->op = OP_NONE;
        ++h->gen;
        pthread_cond_broadcast(&h->cv_done);
    }
    // withdraw the team (what smoqy_team_unserve does)
    h->shutdown = 1;
    snprintf(h->err, sizeof(h->err), "the team was withdrawn by its serving process");
    for (int w = 0; w < K; ++w) mem[w].rc = 10;
    ++h->gen;
    pthread_cond_broadcast(&h->cv_done);
    pthread_mutex_unlock(&h->m);
    usleep(200000);  // members still asleep read their answer before the name goes away (the mapping itself outlives the unlink)
    shm_unlink(name.c_str());
    return 0;
}
