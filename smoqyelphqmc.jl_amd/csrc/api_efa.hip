// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "efa": EFA leapfrog on the device and the HMC trajectory (polling and asynchronous forms).
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- EFA leapfrog on the device -----------------------------------------------------------
// SmoQyDQMC's ExactFourierAccelerator (initialize_momentum!, evolve_eom!, kinetic_energy; call sites src/EFAPFFHMCUpdater.jl:142, 150,
// 202, 244) is NOT under /root/reference: what is built here is fixed by those call sites and by the published algorithm — exact
// harmonic evolution of every τ-Fourier mode of the phonon fields under the quadratic bosonic action, with a per-mode dynamical mass —
// and takes the per-(ω, mode) action eigenvalues q and masses m as INPUTS, so the shim passes whatever its accelerator holds.
// Parity unpinned.

static int efa_launch(smoqy_ctx *c, int mode, double dt, double kick, bool with_force)
{
    auto &F = c->force;
    const Geometry &g = c->g;
    EfaArgs e{};
    e.Lt = g.Lt; e.Nph = F.Nph; e.nw = g.nw; e.SB = F.efa_SB; e.ntile = F.efa_ntile; e.nfac = c->tf.nfac;
    for (int f = 0; f < 16; ++f) e.fac[f] = c->tf.fac[f];
    e.wtab = c->d_wtab;
    e.x = F.d_x; e.p = F.d_p; e.force = with_force ? F.d_out : nullptr; e.kick = kick;
    e.q = F.d_q; e.m = F.d_m; e.finite_mass = F.d_fm; e.dt = dt; e.mode = mode; e.part = F.d_part;
    launch_efa(c->stream, e);
    return check_launch(c, "efa");
}

// (K, S_b) of the state the last efa_launch left, summed over the tiles in fixed order
static int efa_read_energies(smoqy_ctx *c, double *K, double *Sb)
{
    auto &F = c->force;
    const size_t n = 2 * (size_t)c->g.nw * F.efa_ntile;
    HIPCHK(c, hipMemcpyAsync(F.h_part, F.d_part, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int w = 0; w < c->g.nw; ++w) {
        double k = 0.0, s = 0.0;
        for (int t = 0; t < F.efa_ntile; ++t) { k += F.h_part[2 * ((size_t)w * F.efa_ntile + t)]; s += F.h_part[2 * ((size_t)w * F.efa_ntile + t) + 1]; }
        if (K) K[w] = k;
        if (Sb) Sb[w] = s;
    }
    return 0;
}


int smoqy_efa_config(smoqy_ctx *c, const double *q, const double *m)
{
    CHECK_CTX(c);
    auto &F = c->force;
    if (!F.set) FAIL(c, 1, "call smoqy_force_set_couplings first");
    if (!q || !m) FAIL(c, 1, "q and m must be given");
    if (!c->tf_ok) FAIL(c, 5, "the EFA kernels need a time extent that factors into 2, 3, 5, 7 (Ltau = %d)", c->g.Lt);
    const Geometry &g = c->g;
    const size_t nqm = (size_t)g.Lt * std::max(F.Nph, 1), nx = (size_t)g.nw * nqm;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (void *z : {(void *)F.d_p, (void *)F.d_x0, (void *)F.d_q, (void *)F.d_m, (void *)F.d_part, (void *)F.d_fm})
        if (z) (void)hipFree(z);
    if (F.h_part) (void)hipHostFree(F.h_part);
    F.d_p = F.d_x0 = F.d_q = F.d_m = F.d_part = F.h_part = nullptr; F.d_fm = nullptr; F.efa_set = false;
    F.efa_SB = 8;
    while (F.efa_SB > 1 && (2 * (size_t)g.Lt * F.efa_SB + g.Lt) * sizeof(double2) > 150 * 1024) F.efa_SB /= 2;
    F.efa_ntile = (std::max(F.Nph, 1) + F.efa_SB - 1) / F.efa_SB;
    HIPCHK(c, hipMalloc(&F.d_p, nx * sizeof(double)));
    HIPCHK(c, hipMemset(F.d_p, 0, nx * sizeof(double)));
    HIPCHK(c, hipMalloc(&F.d_x0, nx * sizeof(double)));
    HIPCHK(c, hipMemset(F.d_x0, 0, nx * sizeof(double)));
    F.x0_valid = false;
    HIPCHK(c, hipMalloc(&F.d_q, nqm * sizeof(double)));
    HIPCHK(c, hipMalloc(&F.d_m, nqm * sizeof(double)));
    HIPCHK(c, hipMalloc(&F.d_fm, F.finite_mass.size() * sizeof(int)));
    HIPCHK(c, hipMalloc(&F.d_part, 2 * (size_t)g.nw * F.efa_ntile * sizeof(double)));
    HIPCHK(c, hipHostMalloc(&F.h_part, 2 * (size_t)g.nw * F.efa_ntile * sizeof(double)));
    // the ω ↔ −ω symmetry of q and m is what keeps the evolved fields real: check instead of assuming
    for (int p = 0; p < F.Nph; ++p)
        for (int om = 1; om < g.Lt; ++om) {
            const double a = q[p + (size_t)F.Nph * om], b = q[p + (size_t)F.Nph * (g.Lt - om)], ma = m[p + (size_t)F.Nph * om], mb = m[p + (size_t)F.Nph * (g.Lt - om)];
            if (std::fabs(a - b) > 1e-12 * (std::fabs(a) + std::fabs(b)) || (std::isfinite(ma) && std::fabs(ma - mb) > 1e-12 * (std::fabs(ma) + std::fabs(mb))))
                FAIL(c, 1, "q / m of phonon %d are not symmetric under omega -> Ltau - omega (omega = %d)", p + 1, om);
        }
    HIPCHK(c, hipMemcpy(F.d_q, q, nqm * sizeof(double), hipMemcpyHostToDevice));  // Nph x Ltau column-major == [ω][p]
    HIPCHK(c, hipMemcpy(F.d_m, m, nqm * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(F.d_fm, F.finite_mass.data(), F.finite_mass.size() * sizeof(int), hipMemcpyHostToDevice));
    F.efa_set = true;
    return 0;
}

int smoqy_efa_set_state(smoqy_ctx *c, const double *x_all, const double *p_all)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (p_all && nx) {
        HIPCHK(c, hipMemcpyAsync(c->force.d_p, p_all, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (x_all) return smoqy_update_from_phonons_all(c, x_all);  // x and the fields that follow from it
    return 0;
}

int smoqy_efa_get_state(smoqy_ctx *c, double *x_all, double *p_all)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (x_all && nx) HIPCHK(c, hipMemcpyAsync(x_all, c->force.d_x, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (p_all && nx) HIPCHK(c, hipMemcpyAsync(p_all, c->force.d_p, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// fields (exp(-ΔτV), cosh/sinh, Λ) from the device-resident x: the tail of smoqy_update_from_phonons_all without the upload
static int refresh_from_device_x(smoqy_ctx *c)
{
    auto &F = c->force;
    const Geometry &g = c->g;
    ForceArgs a = force_args(c, 0.0, nullptr, nullptr);
    const bool do_t = g.Nh > 0 && (F.Nssh > 0 || !F.t_done);
    launch_phonon_fields(c->stream, a, F.d_bare, F.d_bare + g.N, c->d_expV, c->d_ch, c->d_sh, c->d_lam, g.is_sym ? F.dtau / 2 : F.dtau, do_t, g.is_cplx ? F.d_bare + g.N + g.Nh : nullptr,
                         c->d_shi);
    if (do_t) {
        launch_pack_csf(c->stream, c->d_ch, c->d_sh, c->d_psrc, c->d_csf, c->d_cs_varies, g.nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal, c->d_csi ? c->d_shi : nullptr, c->d_csi);
        for (int w = 0; w < g.nw; ++w) set_cs_const(c, w, F.Nssh == 0 ? F.t0_level : 0);  // no SSH coupling: t is the bare per-bond hopping on every slice
    }
    F.t_done = true;
    return check_launch(c, "refresh_from_device_x");
}

int smoqy_efa_initialize_momentum(smoqy_ctx *c, const double *R_all, double *K)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    if (!R_all) FAIL(c, 1, "R_all is NULL");
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (nx) HIPCHK(c, hipMemcpyAsync(c->force.d_p, R_all, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (int rc = efa_launch(c, 1, 0.0, 0.0, false)) return rc;
    return efa_read_energies(c, K, nullptr);
}

int smoqy_efa_energies(smoqy_ctx *c, double *K, double *Sb)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    if (int rc = efa_launch(c, 2, 0.0, 0.0, false)) return rc;
    return efa_read_energies(c, K, Sb);
}

int smoqy_efa_evolve(smoqy_ctx *c, double dt, double kick_dt, int refresh_fields)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    if (int rc = efa_launch(c, 0, dt, kick_dt, kick_dt != 0.0)) return rc;
    if (refresh_fields) if (int rc = refresh_from_device_x(c)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smoqy_efa_checkpoint(smoqy_ctx *c, int restore)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (!restore) {
        if (nx) HIPCHK(c, hipMemcpyAsync(c->force.d_x0, c->force.d_x, nx * sizeof(double), hipMemcpyDeviceToDevice, c->stream));  // copyto!(x0, x), :130
        c->force.x0_valid = true;
    } else {
        if (!c->force.x0_valid) FAIL(c, 1, "smoqy_efa_checkpoint(ctx, 1): there is no checkpoint to restore (call smoqy_efa_checkpoint(ctx, 0) first)");
        if (nx) HIPCHK(c, hipMemcpyAsync(c->force.d_x, c->force.d_x0, nx * sizeof(double), hipMemcpyDeviceToDevice, c->stream));  // copyto!(x, x0) + update!, :266-275
        if (int rc = refresh_from_device_x(c)) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// the reject branch (:263-275) for SOME walkers of the batch: restore[w] != 0 puts walker w's x back to its checkpoint, the others keep
// the fields the trajectory left (each replica takes its own Metropolis decision)
int smoqy_efa_restore_walkers(smoqy_ctx *c, const int *restore)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    if (!restore) FAIL(c, 1, "restore is NULL");
    if (!c->force.x0_valid) FAIL(c, 1, "smoqy_efa_restore_walkers: there is no checkpoint to restore (call smoqy_efa_checkpoint(ctx, 0) first)");
    const size_t slab = (size_t)c->g.Lt * c->force.Nph;
    bool any = false;
    for (int w = 0; w < c->g.nw; ++w)
        if (restore[w] && slab) {
            HIPCHK(c, hipMemcpyAsync(c->force.d_x + (size_t)w * slab, c->force.d_x0 + (size_t)w * slab, slab * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            any = true;
        }
    if (any)
        if (int rc = refresh_from_device_x(c)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// 0: smoqy_hmc_trajectory_v waits for every force solve (the polling form); 1 (default): the asynchronous form where a previous trajectory
// of the same length and tolerance has left iteration counts to launch on.  runs / misses (may be NULL): asynchronous trajectories so far
// and how many of them had to be repeated with polls.
int smoqy_hmc_async(smoqy_ctx *c, int on, long *runs, long *misses)
{
    CHECK_CTX(c);
    if (on >= 0) c->traj_async = on ? 1 : 0;
    if (runs) *runs = c->traj_async_runs;
    if (misses) *misses = c->traj_async_misses;
    return 0;
}

int smoqy_hmc_trajectory_v(smoqy_ctx *c, int phi, int psi, int Nt, double dt, double tol_force, int maxiter, int use_precond, const double *randvecs, double *Sf, int *iters, double *eps)
{
    CHECK_CTX(c);
    CHECK_EFA(c);
    if (int rc = check_vec(c, phi)) return rc;
    if (int rc = check_vec(c, psi)) return rc;
    if (phi == psi) FAIL(c, 1, "phi and psi must be different vectors");
    const Geometry &g = c->g;
    if (g.nrhs != 1) FAIL(c, 1, "smoqy_hmc_trajectory_v needs a handle with nrhs = 1");
    if (Nt < 1) FAIL(c, 1, "Nt < 1");
    if (use_precond && !randvecs) FAIL(c, 1, "randvecs is NULL");
    std::vector<int> it((size_t)g.nw);
    std::vector<double> ep((size_t)g.nw);
    // S_f of every step stays in its own device slot and comes to the host in one transfer behind the last step: no host synchronisation
    // per step besides the solve's own (round 2 synchronised here 24 times per trajectory) and no copy command between the force kernels
    // and the leapfrog step either
    if ((size_t)Nt * g.nsys > c->traj_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_traj_dot) (void)hipHostFree(c->h_traj_dot);
        if (c->d_traj_dot) (void)hipFree(c->d_traj_dot);
        c->h_traj_dot = nullptr;
        c->d_traj_dot = nullptr;
        c->traj_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_traj_dot, (size_t)Nt * g.nsys * sizeof(double2), hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void **)&c->d_traj_dot, (size_t)Nt * g.nsys * sizeof(double2)));
        c->traj_cap = (size_t)Nt * g.nsys;
    }
    double2 *hdot = c->h_traj_dot;
    // the Lanczos start vectors of all Nt steps cross the boundary once, in front of the trajectory
    const size_t rvn = (size_t)g.nw * g.N * (g.is_cplx ? 2 : 1);
    if (use_precond) {
        if ((size_t)Nt * rvn > c->rand_traj_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));  // a queued kernel may still read the old buffer
            if (c->d_rand_traj) (void)hipFree(c->d_rand_traj);
            c->d_rand_traj = nullptr;
            c->rand_traj_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_rand_traj, (size_t)Nt * rvn * sizeof(double)));
            c->rand_traj_cap = (size_t)Nt * rvn;
        }
        if (int rc = pin_h2d(c, c->d_rand_traj, randvecs, (size_t)Nt * rvn * sizeof(double))) return rc;
    }
    // ---- asynchronous form: no host wait between the steps (round 4) ----
    // Along a trajectory the host used to wait for every force solve (a poll of the CG states: 25-35 µs of idle stream per solve, then ~20
    // launches of force, leapfrog, field and preconditioner kernels issued behind an empty queue).  The iteration counts of step t change
    // by at most a step or two from one trajectory to the next, so each solve is launched with the count its step needed LAST time plus
    // a margin (iterations past convergence are early-exit launches), and the whole trajectory is queued without a single host wait.
    // Nothing is taken on trust: the per-step states are read back at the end, and unless EVERY solve converged (done = 1, finite ϵ — a
    // preconditioner whose status record changed under the launches shows up as a poisoned or unconverged solve) x, p and the fields are
    // put back and the polling form below runs the trajectory again.  Results are those of the polling form bit for bit: the same
    // kernels run the same iterations; iterations after `done` never touch a system's state.
    const size_t nxp = (size_t)g.nw * g.Lt * c->force.Nph;
    // When it pays: a poll costs 25-40 us per solve, a miss costs the whole trajectory a second time.  Solves of a few dozen iterations
    // (Holstein lattices: counts move by a step or two between trajectories) gain 1-4 % (L = 16) to 50 % (L = 4); solves of a hundred and
    // more iterations (SSH models at alpha = 1: counts move by tens) gain nothing from the missing polls and miss often — measured before
    // this rule: optical SSH 180 -> 105, bond SSH 125 -> 71 sweeps/s.  So: only below kAsyncMaxIters iterations per solve, and after a
    // miss the next trajectories poll (1, 2, 4 ... 64 of them for consecutive misses).
    constexpr int kAsyncMaxIters = 64;
    bool async = c->traj_async && use_precond && c->traj_hint.size() == (size_t)Nt && c->traj_hint_tol == tol_force;
    for (int t = 0; t < Nt && async; ++t) async = c->traj_hint[(size_t)t] > 0 && c->traj_hint[(size_t)t] <= kAsyncMaxIters;
    if (async && c->traj_skip > 0) { --c->traj_skip; async = false; }
    if (async) {
        if ((size_t)Nt * g.nsys > c->traj_st_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->h_traj_st) (void)hipHostFree(c->h_traj_st);
            if (c->d_traj_st) (void)hipFree(c->d_traj_st);
            c->h_traj_st = nullptr; c->d_traj_st = nullptr; c->traj_st_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_traj_st, (size_t)Nt * g.nsys * sizeof(CgState), hipHostMallocDefault));
            HIPCHK(c, hipMalloc((void **)&c->d_traj_st, (size_t)Nt * g.nsys * sizeof(CgState)));
            c->traj_st_cap = (size_t)Nt * g.nsys;
        }
        if (2 * nxp > c->traj_save_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_traj_save) (void)hipFree(c->d_traj_save);
            c->d_traj_save = nullptr; c->traj_save_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_traj_save, std::max<size_t>(2 * nxp, 1) * sizeof(double)));
            c->traj_save_cap = 2 * nxp;
        }
        if (nxp) {
            HIPCHK(c, hipMemcpyAsync(c->d_traj_save, c->force.d_x, nxp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_traj_save + nxp, c->force.d_p, nxp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
        // update_preconditioner! carries state from solve to solve (the accepted bounds decide whether order / coefficients are rebuilt,
        // src/KPMPreconditioner.jl:582): a repeated trajectory must start from the preconditioner this one started from, or it is a
        // different (equally valid) sequence of preconditioners and its iterates differ at the level of the tolerance
        if (int rc0 = pstat_wait(c)) return rc0;
        const size_t pre_b[6] = {(size_t)g.nw * 2 * sizeof(double), (size_t)g.nw * sizeof(int), (size_t)g.nw * c->nslot * sizeof(int), (size_t)g.nw * c->nslot * c->maxorder * sizeof(double2),
                                 (size_t)g.nw * 4 * sizeof(int), (size_t)g.nw * sizeof(int)};
        void *const pre_p[6] = {c->d_bounds, c->d_active, c->d_order, c->d_coefs, c->d_pstat, c->d_rebuild};
        size_t pre_off[7] = {0};
        for (int q = 0; q < 6; ++q) pre_off[q + 1] = pre_off[q] + ((pre_b[q] + 255) / 256) * 256;
        if (pre_off[6] > c->traj_pre_cap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->d_traj_pre) (void)hipFree(c->d_traj_pre);
            c->d_traj_pre = nullptr; c->traj_pre_cap = 0;
            HIPCHK(c, hipMalloc((void **)&c->d_traj_pre, pre_off[6]));
            c->traj_pre_cap = pre_off[6];
        }
        for (int q = 0; q < 6; ++q) HIPCHK(c, hipMemcpyAsync(c->d_traj_pre + pre_off[q], pre_p[q], pre_b[q], hipMemcpyDeviceToDevice, c->stream));
        std::vector<int> pre_active((size_t)g.nw), pre_hstat(c->h_pstat, c->h_pstat + (size_t)g.nw * 4);
        for (int w = 0; w < g.nw; ++w) pre_active[(size_t)w] = c->pre[w].active;
        const int pre_heavy = c->cheb_heavy;
        const bool pre_stale = c->mirrors_stale;
        const int pre_maxorder = c->maxorder;
        int rc = efa_launch(c, 0, 0.5 * dt, 0.0, false);
        if (!rc) rc = refresh_from_device_x(c);
        for (int t = 0; t < Nt && !rc; ++t) {
            rc = pff_core(c, phi, psi, nullptr, tol_force, maxiter, use_precond, true, nullptr, nullptr, c->d_rand_traj + (size_t)t * rvn, c->d_traj_dot + (size_t)t * g.nsys, t);
            if (!rc) rc = efa_launch(c, 0, (t == Nt - 1) ? 0.5 * dt : dt, dt, true);
            if (!rc) rc = refresh_from_device_x(c);
        }
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->h_traj_st, c->d_traj_st, (size_t)Nt * g.nsys * sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hdot, c->d_traj_dot, (size_t)Nt * g.nsys * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (int rc2 = check_launch(c, "hmc_trajectory (asynchronous)")) return rc2;
        ++c->traj_async_runs;
        bool ok = true;
        for (size_t k = 0; k < (size_t)Nt * g.nsys && ok; ++k) ok = c->h_traj_st[k].done == 1 && std::isfinite(c->h_traj_st[k].eps);
        if (ok) {
            for (int t = 0; t < Nt; ++t) {
                int mx = 0;
                for (int w = 0; w < g.nw; ++w) {
                    const CgState &st = c->h_traj_st[(size_t)t * g.nsys + w];
                    mx = std::max(mx, st.iters);
                    if (iters) iters[(size_t)t * g.nw + w] = st.iters;
                    if (eps) eps[(size_t)t * g.nw + w] = st.eps;
                    if (Sf) Sf[(size_t)t * g.nw + w] = hdot[(size_t)t * g.nsys + w].x;
                }
                c->traj_hint[(size_t)t] = mx;
            }
            c->traj_backoff /= 2;
            return 0;
        }
        // a solve did not converge within what was launched (or its preconditioner changed under it): back to the start, with polls
        ++c->traj_async_misses;
        c->traj_margin = std::min(c->traj_margin + 2, 16);
        c->traj_backoff = std::min(std::max(1, 2 * c->traj_backoff), 64);
        c->traj_skip = c->traj_backoff;
        if (nxp) {
            HIPCHK(c, hipMemcpyAsync(c->force.d_x, c->d_traj_save, nxp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->force.d_p, c->d_traj_save + nxp, nxp * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
        if (int rc2 = refresh_from_device_x(c)) return rc2;
        if (int rc2 = pstat_wait(c)) return rc2;
        if (c->maxorder != pre_maxorder) FAIL(c, 7, "internal: the preconditioner's coefficient table was resized inside an asynchronous trajectory");
        for (int q = 0; q < 6; ++q) HIPCHK(c, hipMemcpyAsync(pre_p[q], c->d_traj_pre + pre_off[q], pre_b[q], hipMemcpyDeviceToDevice, c->stream));
        std::memcpy(c->h_pstat, pre_hstat.data(), pre_hstat.size() * sizeof(int));
        for (int w = 0; w < g.nw; ++w) c->pre[w].active = pre_active[(size_t)w];
        if (c->cheb_heavy != pre_heavy) { c->cheb_heavy = pre_heavy; drop_graphs(c); }
        c->mirrors_stale = pre_stale || c->mirrors_stale;
    }
    // evolve_eom!(x, p, Δt/2); update!(fdm)                                                            EFAPFFHMCUpdater.jl:148-152
    if (int rc = efa_launch(c, 0, 0.5 * dt, 0.0, false)) return rc;
    if (int rc = refresh_from_device_x(c)) return rc;
    c->traj_hint.assign((size_t)Nt, 0);
    c->traj_hint_tol = tol_force;
    for (int t = 0; t < Nt; ++t) {                                                                   // :162
        const double *d_rv = use_precond ? c->d_rand_traj + (size_t)t * rvn : nullptr;
        if (int rc = pff_core(c, phi, psi, nullptr, tol_force, maxiter, use_precond, true, it.data(), ep.data(), d_rv, c->d_traj_dot + (size_t)t * g.nsys)) return rc;  // :172 (force stays in force.d_out)
        // p -= Δt ∂S/∂x (:196) fused into evolve_eom!(x, p, Δt′) (:201-202); update!(fdm) (:204-205)
        if (int rc = efa_launch(c, 0, (t == Nt - 1) ? 0.5 * dt : dt, dt, true)) return rc;
        if (int rc = refresh_from_device_x(c)) return rc;
        for (int w = 0; w < g.nw; ++w) {
            if (iters) iters[(size_t)t * g.nw + w] = it[w];
            if (eps) eps[(size_t)t * g.nw + w] = ep[w];
            c->traj_hint[(size_t)t] = std::max(c->traj_hint[(size_t)t], it[w]);   // what the next trajectory's asynchronous form launches
        }
    }
    HIPCHK(c, hipMemcpyAsync(hdot, c->d_traj_dot, (size_t)Nt * g.nsys * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (Sf)
        for (int t = 0; t < Nt; ++t)
            for (int w = 0; w < g.nw; ++w) Sf[(size_t)t * g.nw + w] = hdot[(size_t)t * g.nsys + w].x;
    return check_launch(c, "hmc_trajectory");
}


}  // extern "C"
