// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "precond": KPM preconditioner: configuration, update_preconditioner! on the device and its status records, ldiv!.
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- KPM preconditioner ---------------------------------------------------------------------------

int smoqy_precond_config(smoqy_ctx *c, double rbuf, int n_lanczos, double a1, double a2)
{
    CHECK_CTX(c);
    if (n_lanczos < 2 || n_lanczos > 1024 || !(rbuf > 0) || !(a1 > 0) || !(a2 >= 0)) FAIL(c, 1, "invalid preconditioner configuration");
    c->rbuf = rbuf; c->nlanczos = n_lanczos; c->a1 = a1; c->a2 = a2;
    // a1 / a2 set the largest order an active preconditioner can reach: resize the coefficient table and forget the expansions, so that the
    // next update rebuilds them with the new parameters
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int stride = std::max(64, coef_table_stride(c));
    if (stride != c->maxorder) {
        drop_graphs(c);
        HIPCHK(c, hipFree(c->d_coefs));
        c->d_coefs = nullptr;
        HIPCHK(c, hipMalloc(&c->d_coefs, (size_t)c->g.nw * c->nslot * stride * sizeof(double2)));
        c->maxorder = stride;
    }
    HIPCHK(c, hipMemset(c->d_coefs, 0, (size_t)c->g.nw * c->nslot * c->maxorder * sizeof(double2)));
    HIPCHK(c, hipMemset(c->d_bounds, 0, (size_t)c->g.nw * 2 * sizeof(double)));
    HIPCHK(c, hipMemset(c->d_active, 0, (size_t)c->g.nw * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_order, 0, (size_t)c->g.nw * c->nslot * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_pstat, 0, (size_t)c->g.nw * 4 * sizeof(int)));
    for (auto &p : c->pre) { p.active = 0; p.emin = p.emax = 0.0; std::fill(p.order.begin(), p.order.end(), 0); for (auto &v : p.coefs) v.clear(); }
    // the page-locked mirror of the status records too: a later update of ONE walker consumes the records of ALL walkers, and stale
    // "active" / heavy counts of the untouched ones would come back
    std::memset(c->h_pstat, 0, (size_t)c->g.nw * 4 * sizeof(int));
    c->pstat_pending = false;
    c->pstat_ever = false;
    c->mirrors_stale = false;
    c->cheb_heavy = 0;
    return 0;
}

static void pstat_consume(smoqy_ctx *c);
static int refresh_mirrors(smoqy_ctx *c);

// host-supplied preconditioner state of one walker (smoqy_precond_set) -> device tables and status record
static int upload_precond(smoqy_ctx *c, int w)
{
    const WalkerPrecond &p = c->pre[w];
    int need = 1;
    for (int o : p.order) need = std::max(need, o);
    if (need > c->maxorder) {  // grow the padded coefficient table and re-upload every walker
        int cap = c->maxorder;
        while (cap < need) cap *= 2;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        drop_graphs(c);  // captured Chebyshev launches hold the old table pointer and stride
        HIPCHK(c, hipFree(c->d_coefs));
        c->d_coefs = nullptr;
        HIPCHK(c, hipMalloc(&c->d_coefs, (size_t)c->g.nw * c->nslot * cap * sizeof(double2)));
        c->maxorder = cap;
        for (int ww = 0; ww < c->g.nw; ++ww)
            if (ww != w && !c->pre[ww].order.empty()) if (int rc = upload_precond(c, ww)) return rc;
    }
    std::vector<double2> tab((size_t)c->nslot * c->maxorder, make_double2(0.0, 0.0));
    for (int s = 0; s < c->nslot; ++s)
        for (size_t k = 0; k < p.coefs[s].size(); ++k) tab[(size_t)s * c->maxorder + k] = p.coefs[s][k];
    const double bnd[2] = {p.emin, p.emax};
    // the host sides are temporaries (a local table, a stack pair, members of a vector that may be reallocated): through the page-locked arena
    if (int rc = pin_h2d(c, c->d_coefs + (size_t)w * c->nslot * c->maxorder, tab.data(), tab.size() * sizeof(double2))) return rc;
    if (int rc = pin_h2d(c, c->d_order + (size_t)w * c->nslot, p.order.data(), (size_t)c->nslot * sizeof(int))) return rc;
    if (int rc = pin_h2d(c, c->d_bounds + 2 * (size_t)w, bnd, sizeof(bnd))) return rc;
    if (int rc = pin_h2d(c, c->d_active + w, &p.active, sizeof(int))) return rc;
    {   // this walker's status record, as the device bookkeeping would have written it: how many leading ranks (rank 2s and 2s+1 share
        // slot s, KPMPreconditioner.jl:387; Asym: slots l and Lτ-1-l) carry a chain — the light workgroups of cheb_own_kernel take
        // everything behind them
        int last = -1;
        for (int sl = 0; sl < (int)p.order.size(); ++sl)
            if (p.order[sl] > 1) last = std::max(last, c->g.is_sym ? sl : std::min(sl, c->g.Lt - 1 - sl));
        int *st = c->h_pstat + 4 * (size_t)w;
        st[0] += 1;
        st[1] = std::min(c->g.Lt, 2 * (last + 1));
        st[2] = need;
        st[3] = p.active;
        if (int rc = pin_h2d(c, c->d_pstat + 4 * (size_t)w, st, 4 * sizeof(int))) return rc;
        const int zero = 0;
        if (int rc = pin_h2d(c, c->d_rebuild + w, &zero, sizeof(int))) return rc;
        pstat_consume(c);
    }
    return 0;
}

// the PreUpd argument of the Lanczos / expansion kernels (kernels_kpm.hip)
static PreUpd pre_upd(smoqy_ctx *c)
{
    PreUpd u{};
    u.bounds = c->d_bounds; u.active = c->d_active; u.order = c->d_order; u.coefs = c->d_coefs; u.rebuild = c->d_rebuild; u.status = c->d_pstat;
    u.rbuf = c->rbuf; u.a1 = c->g.is_sym ? 2.0 * c->a1 : c->a1; u.a2 = c->a2;  // :263
    u.nslot = c->nslot; u.maxorder = c->maxorder; u.Lt = c->g.Lt; u.is_sym = c->g.is_sym;
    return u;
}

// The status records of the last update_preconditioner! have landed: refresh what the host keeps of them — the activation flags (they
// choose the CG path) and the count of leading frequencies with a multi-term expansion (the launch geometry of cheb_own_kernel).
static void pstat_consume(smoqy_ctx *c)
{
    int last = 0;
    for (int w = 0; w < c->g.nw; ++w) {
        const int *st = c->h_pstat + 4 * (size_t)w;
        c->pre[w].active = st[3];
        last = std::max(last, st[1]);
    }
    const int heavy = std::min(c->g.Lt, last);
    if (heavy != c->cheb_heavy) { c->cheb_heavy = heavy; drop_graphs(c); }  // a captured CG graph holds the old count
    c->pstat_ever = true;
}

// Block until the status records of the last update have arrived (no-op when none is outstanding).  The copy sits in the stream right
// behind the bookkeeping kernel, so with other work queued behind it the GPU does not idle while the host wakes up.
int pstat_wait(smoqy_ctx *c)
{
    if (!c->pstat_pending) return 0;
    HIPCHK(c, hipEventSynchronize(c->ev_pstat));
    c->pstat_pending = false;
    pstat_consume(c);
    return 0;
}

// update_preconditioner! (:554-597) for walkers [w0, w0 + nw), entirely on the device and without a host synchronisation: τ-means
// (update_B̄! :604-621), Lanczos from the caller's start vectors (calculate_bounds! :625-658) ending with the tridiagonal extremes, the
// widening, the activation test and the "bounds moved by more than rbuf/2" decision (:569-593), then the expansion coefficients of the
// walkers whose bounds were accepted (:734-795).  A 16-byte status record per walker follows the kernels to the host (pstat_wait).
// d_randvecs: the start vectors are on the device already (a trajectory sends those of all its steps in one transfer) — no copy command
// between the τ-means and the Lanczos kernel, and no arena turnover (a stream drain) every few steps of a large batch.
int precond_update_range(smoqy_ctx *c, int w0, int nw, const double *randvecs, const double *d_randvecs)
{
    const Geometry &g = c->g;
    const int n = c->nlanczos;
    if (int rc = pstat_wait(c)) return rc;  // one outstanding record at a time (h_pstat is about to be overwritten)
    launch_tau_means(c->stream, c->kg, c->d_expV, c->d_ch, c->d_sh, c->d_dbar, c->d_cbar, c->d_sbar, g.Lt, g.N, g.Nh, w0, nw, c->d_shi, c->d_sbari);
    if (!d_randvecs)
        if (int rc = pin_h2d(c, c->d_rand, randvecs, (size_t)nw * g.N * (g.is_cplx ? 2 : 1) * sizeof(double))) return rc;  // complex T: N complex deviates per walker (:634); the caller's array may be pageable
    KpmArgs k = kpm_args(c, nullptr, nullptr);
    const PreUpd u = pre_upd(c);
    launch_lanczos(c->stream, k, c->kg, w0, nw, d_randvecs ? d_randvecs : c->d_rand, n, c->d_lan + (size_t)w0 * 1024, c->d_lan + (size_t)(g.nw + w0) * 1024, !g.is_sym, u);
    launch_kpm_expansions(c->stream, u, w0, nw);
    HIPCHK(c, hipMemcpyAsync(c->h_pstat + 4 * (size_t)w0, c->d_pstat + 4 * (size_t)w0, (size_t)nw * 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_pstat, c->stream));
    c->pstat_pending = true;
    c->mirrors_stale = true;
    return check_launch(c, "precond_update");
}

// host copies of one walker's preconditioner state (smoqy_precond_get*): bounds, order, coefficients, Lanczos coefficients
static int refresh_mirrors(smoqy_ctx *c)
{
    if (int rc = pstat_wait(c)) return rc;
    if (!c->mirrors_stale) return 0;
    const Geometry &g = c->g;
    const int n = c->nlanczos;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<double> bnd((size_t)g.nw * 2), lan((size_t)g.nw * 2 * 1024);
    std::vector<int> ord((size_t)g.nw * c->nslot);
    std::vector<double2> cf((size_t)g.nw * c->nslot * c->maxorder);
    HIPCHK(c, hipMemcpy(bnd.data(), c->d_bounds, bnd.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(lan.data(), c->d_lan, lan.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ord.data(), c->d_order, ord.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cf.data(), c->d_coefs, cf.size() * sizeof(double2), hipMemcpyDeviceToHost));
    for (int w = 0; w < g.nw; ++w) {
        WalkerPrecond &p = c->pre[w];
        p.emin = bnd[2 * (size_t)w];
        p.emax = bnd[2 * (size_t)w + 1];
        p.lan_a.assign(lan.begin() + (size_t)w * 1024, lan.begin() + (size_t)w * 1024 + n);
        p.lan_b.assign(lan.begin() + (size_t)(g.nw + w) * 1024, lan.begin() + (size_t)(g.nw + w) * 1024 + n - 1);
        for (int sl = 0; sl < c->nslot; ++sl) {
            p.order[sl] = ord[(size_t)w * c->nslot + sl];
            const double2 *src = cf.data() + ((size_t)w * c->nslot + sl) * c->maxorder;
            p.coefs[sl].assign(src, src + std::max(p.order[sl], 0));
        }
    }
    c->mirrors_stale = false;
    return 0;
}

int smoqy_matvec_force_generic(smoqy_ctx *c, int on)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    c->ff.enabled = (!on && (!g.is_cplx || g.is_sym) && g.ncol >= 1 && g.ncol <= kFdmColours && c->ff.threads <= 1024) ? 1 : 0;
    drop_graphs(c);
    choose_chunking(c);
    return 0;
}

int smoqy_precond_force_generic(smoqy_ctx *c, int on)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    int maxp = 0;
    (void)g;
    maxp = c->kg.threads;
    c->kg.fast = (!on && !g.is_cplx && g.ncol >= 1 && g.ncol <= kMaxColours && maxp <= 1024) ? 1 : 0;
    c->kg.cplx_fast = (!on && c->d_pcsi) ? 1 : 0;
    drop_graphs(c);
    return 0;
}

int smoqy_precond_update(smoqy_ctx *c, int w, const double *randvec)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    return precond_update_range(c, w, 1, randvec);
}

int smoqy_precond_update_all(smoqy_ctx *c, const double *randvecs)
{
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    return precond_update_range(c, 0, c->g.nw, randvecs);
}

int smoqy_precond_get(smoqy_ctx *c, int w, int *active, double *bounds, int *order, int *norder, double *la, double *lb)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = refresh_mirrors(c)) return rc;  // the state lives on the device since round 3: fetched on demand
    const WalkerPrecond &p = c->pre[w];
    if (active) *active = p.active;
    if (bounds) { bounds[0] = p.emin; bounds[1] = p.emax; }
    if (order) std::copy(p.order.begin(), p.order.end(), order);
    if (norder) *norder = c->nslot;
    if (la) std::copy(p.lan_a.begin(), p.lan_a.end(), la);
    if (lb) std::copy(p.lan_b.begin(), p.lan_b.end(), lb);
    return 0;
}

int smoqy_precond_get_coefs(smoqy_ctx *c, int w, int slot, void *coefs)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    if (slot < 0 || slot >= c->nslot) FAIL(c, 1, "slot %d out of range", slot);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = refresh_mirrors(c)) return rc;
    const auto &v = c->pre[w].coefs[slot];
    std::memcpy(coefs, v.data(), v.size() * sizeof(double2));
    return 0;
}

int smoqy_precond_set(smoqy_ctx *c, int w, int active, const double *bounds, const int *order, const void *coefs)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    const Geometry &g = c->g;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = refresh_mirrors(c)) return rc;  // the other walkers' mirrors must be current: a table growth re-uploads them
    WalkerPrecond &p = c->pre[w];
    p.active = active ? 1 : 0;
    p.emin = bounds[0];
    p.emax = bounds[1];
    const double2 *src = (const double2 *)coefs;
    for (int s = 0; s < c->nslot; ++s) {
        if (order[s] < 1) FAIL(c, 1, "order[%d] = %d < 1", s, order[s]);
        p.order[s] = order[s];
        p.coefs[s].assign(src, src + order[s]);
        src += order[s];
    }
    launch_tau_means(c->stream, c->kg, c->d_expV, c->d_ch, c->d_sh, c->d_dbar, c->d_cbar, c->d_sbar, g.Lt, g.N, g.Nh, w, 1, c->d_shi, c->d_sbari);
    return upload_precond(c, w);
}

// frequency-space part of ldiv!(u', P, u): v = FFT⁻¹ · (per-ω Chebyshev / Lτ) · FFT src, in the
// twiddled basis (the θ phases are the caller's business).  part_rz, when given, receives the
// Parseval partials of src·v per (system, ω).
int precond_core(smoqy_ctx *c, const double2 *src, double2 *v, const CgState *cg, double2 *part_rz, bool half)
{
    const bool own = c->tf_ok && c->use_tfft;
    void *in[1] = {(void *)src}, *out[1] = {v};
    if (int rc = pstat_wait(c)) return rc;  // the Chebyshev launch below takes its geometry from the last update's status record
    if (own) {
        TfftArgs t = c->tf;
        t.src = src; t.dst = v; t.pre_tw = nullptr; t.post_tw = nullptr;
        launch_tfft(c->stream, 0, t);                                                   // KPMPreconditioner.jl:375
    } else if (src == v) FFTCHK(c, rocfft_execute(c->plan_f, out, nullptr, c->fft_info));
    else FFTCHK(c, rocfft_execute(c->plan_f_oop, in, out, c->fft_info));
    KpmArgs k = kpm_args(c, v, cg);
    k.part_rz = part_rz;
    k.half = half ? 1 : 0;
    launch_cheb(c->stream, k, c->kg);                                                   // :381-400 (no transposes needed in this layout)
    c->cheb_name = cheb_kernel_name(k, c->kg);
    if (half) launch_conj_mirror(c->stream, v, c->g.Lt, c->g.N, c->g.nsys);             // :334 / :468
    if (own) {
        TfftArgs t = c->tf;
        t.src = v; t.dst = v; t.pre_tw = nullptr; t.post_tw = nullptr;
        launch_tfft(c->stream, 1, t);                                                   // :406
    } else FFTCHK(c, rocfft_execute(c->plan_b, out, nullptr, c->fft_info));
    return check_launch(c, "precond_core");
}

static int precond_apply_dev(smoqy_ctx *c, double2 *out, const double2 *in, bool half = false)
{
    const Geometry &g = c->g;
    // walkers with an inactive preconditioner come out as the identity (:410) — the Chebyshev
    // kernel reduces to the 1/Lτ scale for them
    HIPCHK(c, hipMemcpyAsync(c->cg_v, in, c->vec_elems() * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    launch_fft_twiddle(c->stream, c->cg_v, c->d_th, g.Lt, g.N, g.nsys, 0);   // θ  (FourierTransformer.jl:46; the 1/√Lτ pair is in the kernel's scale)
    if (int rc = precond_core(c, c->cg_v, c->cg_v, nullptr, nullptr, half)) return rc;
    launch_fft_twiddle(c->stream, c->cg_v, c->d_th, g.Lt, g.N, g.nsys, 1);   // θ⁻¹ (:61)
    HIPCHK(c, hipMemcpyAsync(out, c->cg_v, c->vec_elems() * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    return check_launch(c, "precond_apply");
}

int smoqy_precond_apply_v(smoqy_ctx *c, int out, int in)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, out)) return rc;
    if (int rc = check_vec(c, in)) return rc;
    return precond_apply_dev(c, c->vecs[out], c->vecs[in]);
}

int smoqy_precond_apply(smoqy_ctx *c, void *out, const void *in, int sys0, int count)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_into(c, c->scr[1], in, sys0, count)) return rc;
    if (int rc = precond_apply_dev(c, c->scr[2], c->scr[1])) return rc;
    return download_from(c, c->scr[2], out, sys0, count);
}

// ldiv!(u′, P, u) for REAL vectors (Sym KPMPreconditioner.jl:288-352, Asym :417-485): u is promoted to complex (:306), only the
// frequencies ω < cld(Lτ, 2) go through the Chebyshev kernels, the other half is filled in as their complex conjugate (:334) and
// the real part of the back-transform is returned (:344).  Inactive preconditioner: copy (:349).
int smoqy_precond_apply_real(smoqy_ctx *c, double *out, const double *in, int sys0, int count)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    const Geometry &g = c->g;
    const size_t n = (size_t)count * g.Lt * g.N;
    if (int rc = ensure_stage_real(c, n)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_stage_real, in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_real_to_complex(c->stream, c->d_stage_real, c->d_stage, n);
    launch_transpose_in(c->stream, c->d_stage, c->scr[1], g.Lt, g.N, g.nsys, sys0, count);
    if (int rc = precond_apply_dev(c, c->scr[2], c->scr[1], true)) return rc;
    launch_transpose_out(c->stream, c->scr[2], c->d_stage, g.Lt, g.N, g.nsys, sys0, count);
    launch_complex_to_real(c->stream, c->d_stage, c->d_stage_real, n);
    HIPCHK(c, hipMemcpyAsync(out, c->d_stage_real, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "precond_apply_real");
}


}  // extern "C"
