// Member side of a published walker team (include/smoqy_hip.h, "walker teams" / smoqy_member_*): what a rank of the reference's
// one-walker-per-rank model links when its walker lives on another process's batched handle (tutorials/holstein_honeycomb_mpi.jl:60-72).
// It copies its arrays into the team's shared-memory segment, deposits its scalars and sleeps until the serving process has run the
// round.  No GPU, no HIP runtime, no rocFFT: this file is part of libsmoqy_hip.so AND is built alone as libsmoqy_member.so
// (g++ -shared -pthread -lrt), so a GPU-less rank does not have to load the ROCm libraries.
#include <chrono>
#include <string>

#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/smoqy_hip.h"
#include "team_shm.h"

using namespace smoqy_team_detail;

struct smoqy_member {
    ShmHeader *h = nullptr;
    int w = -1;
    std::string err;
};

static std::string g_member_error;

extern "C" {

const char *smoqy_member_last_error(const smoqy_member *m) { return m ? m->err.c_str() : g_member_error.c_str(); }

int smoqy_member_attach(smoqy_member **out, const char *name, int w, double wait_seconds)
{
    if (!out || !name) { g_member_error = "smoqy_member_attach: null argument"; return 1; }
    *out = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto waited = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    ShmHeader *h = nullptr;
    for (;;) {  // the ranks of a job start together: the serving rank may not have published yet
        const int fd = shm_open(name, O_RDWR, 0600);
        if (fd >= 0) {
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(ShmHeader)) {
                void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (p != MAP_FAILED) {
                    ShmHeader *q = (ShmHeader *)p;
                    if (__atomic_load_n(&q->magic, __ATOMIC_ACQUIRE) == kShmMagic && q->total == (size_t)st.st_size) { h = q; close(fd); break; }
                    munmap(p, (size_t)st.st_size);
                }
            }
            close(fd);
        }
        if (waited() >= wait_seconds) { g_member_error = std::string("smoqy_member_attach: no team published as ") + name; return 9; }
        usleep(2000);
    }
    if (w < 0 || w >= h->K) { g_member_error = "smoqy_member_attach: member index outside the team"; munmap(h, h->total); return 1; }
    ShmMember *mem = (ShmMember *)((char *)h + h->off_members);
    shm_lock(h);
    // `attached` holds the pid of the rank that owns the index; a rank that died without detaching (kill(pid, 0) says ESRCH) is replaced
    const int owner = mem[w].attached;
    const bool taken = owner != 0 && !(kill((pid_t)owner, 0) != 0 && errno == ESRCH);
    if (!taken) mem[w].attached = (int)getpid();
    pthread_mutex_unlock(&h->m);
    if (taken) { g_member_error = "smoqy_member_attach: this member index is already attached"; munmap(h, h->total); return 1; }
    smoqy_member *m = new smoqy_member();
    m->h = h; m->w = w;
    *out = m;
    return 0;
}

int smoqy_member_detach(smoqy_member *m)
{
    if (!m) return 0;
    if (m->h) {
        ShmMember *mem = (ShmMember *)((char *)m->h + m->h->off_members);
        shm_lock(m->h);
        mem[m->w].attached = 0;
        pthread_mutex_unlock(&m->h->m);
        munmap(m->h, m->h->total);
    }
    delete m;
    return 0;
}

int smoqy_member_dims(const smoqy_member *m, int *dims)
{
    if (!m || !dims) return 1;
    dims[0] = m->h->Lt; dims[1] = m->h->N; dims[2] = m->h->K; dims[3] = m->h->Nph;
    return 0;
}

int smoqy_member_fields(const smoqy_member *m, double *x)
{
    if (!m || !x) return 1;
    const size_t nx = (size_t)std::max(m->h->Nph, 1) * m->h->Lt;
    std::memcpy(x, (const char *)m->h + m->h->off_x + (size_t)m->w * nx * sizeof(double), nx * sizeof(double));
    return 0;
}

}  // extern "C"

static int member_round(smoqy_member *m, int op, const Slot &a)
{
    if (!m || !m->h) return 1;
    ShmHeader *h = m->h;
    const int w = m->w;
    char *base = (char *)h;
    Stage g;
    g.R = base + h->off_R; g.x = (double *)(base + h->off_x); g.rv = (double *)(base + h->off_rv); g.dS = (double *)(base + h->off_dS);
    g.P = (double *)(base + h->off_P); g.rvs = (double *)(base + h->off_rvs);
    if (h->ge_Nrv > 0) { g.GR = base + h->off_GR; g.G = base + h->off_G; g.Nrv = h->ge_Nrv; g.gbytes = h->ge_gbytes; }
    stage_in(g, h->K, h->Lt, h->N, h->Nph, w, a);
    ShmMember &q = ((ShmMember *)(base + h->off_members))[w];
    shm_lock(h);
    if (h->shutdown) { m->err = h->err; pthread_mutex_unlock(&h->m); return 10; }
    if (h->arrived > 0 && h->op != op) { m->err = "team members made different calls in the same round"; pthread_mutex_unlock(&h->m); return 8; }
    h->op = op;
    q.has_R = a.R != nullptr; q.has_x = a.x != nullptr; q.has_rv = a.rv != nullptr; q.has_P = a.P != nullptr; q.has_rvs = a.rvs != nullptr;
    q.want_force = a.dSdx != nullptr; q.want_xnew = a.x_new != nullptr;
    q.has_Rrv = a.Rrv != nullptr; q.want_G = a.G != nullptr; q.orb_a = a.orb_a; q.orb_b = a.orb_b;
    q.tol = a.tol; q.maxiter = a.maxiter; q.use_precond = a.use_precond;
    q.Nt = a.Nt; q.dt = a.dt; q.tol_force = a.tol_force; q.accept = a.accept;
    const unsigned long my_gen = h->gen;
    if (++h->arrived == h->K) pthread_cond_signal(&h->cv_arrive);
    // The deadline covers the wait for the OTHER MEMBERS only.  Once the server has taken the round (h->running, set under this lock)
    // its results go through every member's slot and staging part: a member whose deadline passes then keeps waiting for the round's
    // result — in slices, so that a serving process that died inside the round is noticed (code 10) — and is never taken out of
    // `arrived` (leaving mid-round handed the server a slot its owner was already refilling).
    timespec dl = deadline_after(h->timeout_s);
    while (h->gen == my_gen) {
        const int e = pthread_cond_timedwait(&h->cv_done, &h->m, &dl);
        if (e == EOWNERDEAD) pthread_mutex_consistent(&h->m);
        if (e == ETIMEDOUT && h->gen == my_gen) {
            if (h->running) {
                if (h->server_pid > 0 && kill((pid_t)h->server_pid, 0) != 0 && errno == ESRCH) {
                    m->err = "the serving process of this team died inside a round";
                    pthread_mutex_unlock(&h->m);
                    return 10;
                }
                dl = deadline_after(1.0);
                continue;
            }
            --h->arrived;
            m->err = "team rendezvous timed out: not every member made the call";
            pthread_mutex_unlock(&h->m);
            return 9;
        }
    }
    const int rc = q.rc;
    if (rc) m->err = h->err;
    const ShmMember r = q;  // this member's results, copied under the lock
    pthread_mutex_unlock(&h->m);
    if (rc == 0) {
        if (a.Sf) *a.Sf = r.Sf;
        if (a.iters) *a.iters = r.iters;
        if (a.eps) *a.eps = r.eps;
        if (a.RdotR) *a.RdotR = r.RdotR;
        if (a.H0) std::memcpy(a.H0, r.H0, sizeof(r.H0));
        if (a.H1) std::memcpy(a.H1, r.H1, sizeof(r.H1));
        stage_out(g, h->Lt, h->Nph, w, a);
    }
    return rc;
}

extern "C" {

int smoqy_member_sample_phi(smoqy_member *m, const void *R, double *RdotR)
{
    if (!R) return 1;
    Slot s;
    s.R = R; s.RdotR = RdotR;
    return member_round(m, OP_SAMPLE, s);
}

int smoqy_member_pff_step(smoqy_member *m, const double *x, const double *randvec, double tol, int maxiter, int use_precond, double *Sf, int *iters, double *eps, double *dSdx)
{
    Slot s;
    s.x = x; s.rv = randvec; s.tol = tol; s.maxiter = maxiter; s.use_precond = use_precond ? 1 : 0;
    s.Sf = Sf; s.iters = iters; s.eps = eps; s.dSdx = dSdx;
    return member_round(m, OP_PFF, s);
}

int smoqy_member_hmc_update(smoqy_member *m, const double *x, const void *R, const double *P, const double *randvecs, int Nt, double dt, double tol_force, double tol, int maxiter,
                            double *H0, double *H1, double *x_new, int *iters)
{
    if (!R || !P || !randvecs) return 1;
    if (Nt < 1 || Nt > kMaxNt) { if (m) m->err = "smoqy_member_hmc_update: Nt outside 1 … 64"; return 1; }
    Slot s;
    s.x = x; s.R = R; s.P = P; s.rvs = randvecs; s.Nt = Nt; s.dt = dt; s.tol_force = tol_force; s.tol = tol; s.maxiter = maxiter;
    s.H0 = H0; s.H1 = H1; s.x_new = x_new; s.iters = iters;
    return member_round(m, OP_HMC, s);
}

int smoqy_member_hmc_finish(smoqy_member *m, int accept)
{
    Slot s;
    s.accept = accept ? 1 : 0;
    return member_round(m, OP_FINISH, s);
}

int smoqy_member_ge_dims(const smoqy_member *m, int *Nrv, size_t *g_bytes)
{
    if (!m || !m->h) return 1;
    if (Nrv) *Nrv = m->h->ge_Nrv;
    if (g_bytes) *g_bytes = m->h->ge_gbytes;
    return 0;
}

int smoqy_member_ge_update(smoqy_member *m, const void *R, const double *randvec, double tol, int maxiter, int *iters, double *eps)
{
    if (!R || !randvec) return 1;
    Slot s;
    s.Rrv = R; s.rv = randvec; s.tol = tol; s.maxiter = maxiter; s.iters = iters; s.eps = eps;
    return member_round(m, OP_GE_UPDATE, s);
}

int smoqy_member_ge_measure_GD0(smoqy_member *m, int a, int b, void *out)
{
    if (!out) return 1;
    Slot s;
    s.orb_a = a; s.orb_b = b; s.G = out;
    return member_round(m, OP_GE_GD0, s);
}

}  // extern "C"

