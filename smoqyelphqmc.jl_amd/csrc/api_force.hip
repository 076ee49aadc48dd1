// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "force": force terms, device-side update! from the phonon fields, calculate_derivative_fermionic_action! in one call.
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- force terms ----------------------------------------------------------------------------------------

int smoqy_force_set_couplings(smoqy_ctx *c, const smoqy_couplings *cp)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    if (g.nrhs != 1) FAIL(c, 1, "the force entry points need a handle with nrhs = 1");
    if (!cp || cp->Nph < 0 || cp->Nholstein < 0 || cp->Nssh < 0) FAIL(c, 1, "invalid couplings");
    auto &F = c->force;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (void *q : {F.blob, (void *)F.d_x, (void *)F.d_out, (void *)F.d_bare, (void *)F.d_p, (void *)F.d_x0, (void *)F.d_q, (void *)F.d_m, (void *)F.d_part, (void *)F.d_fm})
        if (q) (void)hipFree(q);
    if (F.h_out) (void)hipHostFree(F.h_out);
    if (F.h_part) (void)hipHostFree(F.h_part);
    F = smoqy_ctx::ForceState{};
    const int Nph = cp->Nph, Nhol = cp->Nholstein, Nssh = cp->Nssh;
    const int Q = 2 * Nhol + 2 * Nssh;
    // integer tables
    std::vector<int> h_c2p(Nhol), h_c2s(Nhol), h_ps(Nhol), s_c2p(2 * (size_t)Nssh), bond_ptr((size_t)g.Nh + 1, 0), bond_cpl((size_t)Nssh);
    for (int k = 0; k < Nhol; ++k) {
        const int64_t p = cp->h_coupling_to_phonon[k], i = cp->h_coupling_to_site[k];
        if (p < 1 || p > Nph || i < 1 || i > g.N) FAIL(c, 1, "holstein coupling %d: phonon %lld / site %lld out of range", k + 1, (long long)p, (long long)i);
        h_c2p[k] = (int)p - 1; h_c2s[k] = (int)i - 1; h_ps[k] = cp->h_ph_sym[k] ? 1 : 0;
    }
    for (int k = 0; k < Nssh; ++k) {
        const int64_t p = cp->s_coupling_to_phonon[2 * k], pp = cp->s_coupling_to_phonon[2 * k + 1], n = cp->s_bond[k];
        if (p < 1 || p > Nph || pp < 1 || pp > Nph || n < 1 || n > g.Nh) FAIL(c, 1, "ssh coupling %d: phonons %lld, %lld / bond %lld out of range", k + 1, (long long)p, (long long)pp, (long long)n);
        s_c2p[2 * k] = (int)p - 1; s_c2p[2 * k + 1] = (int)pp - 1;
        bond_ptr[n]++;  // counts, shifted by one
    }
    for (int h = 0; h < g.Nh; ++h) bond_ptr[h + 1] += bond_ptr[h];
    {
        std::vector<int> fill(bond_ptr.begin(), bond_ptr.end() - 1);
        for (int k = 0; k < Nssh; ++k) bond_cpl[fill[(int)cp->s_bond[k] - 1]++] = k;  // coupling order within a bond = ascending c (hopping_to_couplings order)
    }
    // phonon -> contribution slots, in the order the reference adds them: dK pass 0/1 (bond order), dV, dΛ
    std::vector<std::vector<std::pair<int, double>>> lists((size_t)Nph);
    for (int k = 0; k < Nssh; ++k)
        for (int pass = 0; pass < 2; ++pass) {
            const int slot = Nhol + 2 * k + pass;
            if (cp->finite_mass[s_c2p[2 * k]]) lists[s_c2p[2 * k]].push_back({slot, -1.0});          // :229-231
            if (cp->finite_mass[s_c2p[2 * k + 1]]) lists[s_c2p[2 * k + 1]].push_back({slot, +1.0});  // :233-235
        }
    for (int k = 0; k < Nhol; ++k) {
        if (cp->finite_mass[h_c2p[k]]) lists[h_c2p[k]].push_back({k, 1.0});                          // :274
        lists[h_c2p[k]].push_back({Nhol + 2 * Nssh + k, 1.0});                                        // holstein_shift_matrix.jl:193
    }
    std::vector<int> ph_ptr((size_t)Nph + 1, 0), ph_slot;
    std::vector<double> ph_sign;
    for (int p = 0; p < Nph; ++p) {
        for (auto &e : lists[p]) { ph_slot.push_back(e.first); ph_sign.push_back(e.second); }
        ph_ptr[p + 1] = (int)ph_slot.size();
    }
    // site -> Holstein couplings (coupling order within a site, like the reference's loop over c)
    std::vector<int> site_ptr((size_t)g.N + 1, 0), site_cpl((size_t)Nhol);
    for (int k = 0; k < Nhol; ++k) site_ptr[h_c2s[k] + 1]++;
    for (int i = 0; i < g.N; ++i) site_ptr[i + 1] += site_ptr[i];
    {
        std::vector<int> fill(site_ptr.begin(), site_ptr.end() - 1);
        for (int k = 0; k < Nhol; ++k) site_cpl[fill[h_c2s[k]]++] = k;
    }
    // one blob: [ints | doubles]
    const size_t n_int = h_c2p.size() + h_c2s.size() + h_ps.size() + s_c2p.size() + bond_ptr.size() + bond_cpl.size() + ph_ptr.size() + ph_slot.size() + site_ptr.size() + site_cpl.size();
    const size_t n_dbl = 4 * (size_t)Nhol + (g.is_cplx ? 8 : 4) * (size_t)Nssh + ph_sign.size();
    const size_t int_bytes = ((n_int * sizeof(int) + 15) / 16) * 16;
    HIPCHK(c, hipMalloc(&F.blob, int_bytes + n_dbl * sizeof(double) + 16));
    std::vector<int> ib;
    ib.reserve(n_int);
    auto put_i = [&](const std::vector<int> &v) { const size_t off = ib.size(); ib.insert(ib.end(), v.begin(), v.end()); return (const int *)F.blob + off; };
    ForceArgs &t = F.tmpl;
    t = ForceArgs{};
    t.h_c2p = put_i(h_c2p); t.h_c2s = put_i(h_c2s); t.h_phsym = put_i(h_ps); t.s_c2p = put_i(s_c2p);
    t.bond_ptr = put_i(bond_ptr); t.bond_cpl = put_i(bond_cpl); t.ph_ptr = put_i(ph_ptr); t.ph_slot = put_i(ph_slot);
    t.site_ptr = put_i(site_ptr); t.site_cpl = put_i(site_cpl);
    std::vector<double> db;
    db.reserve(n_dbl);
    const double *dbase = (const double *)((const char *)F.blob + int_bytes);
    auto put_d = [&](const double *src, size_t n) { const size_t off = db.size(); db.insert(db.end(), src, src + n); return dbase + off; };
    t.h_alpha = put_d(cp->h_alpha, Nhol); t.h_alpha2 = put_d(cp->h_alpha2, Nhol); t.h_alpha3 = put_d(cp->h_alpha3, Nhol); t.h_alpha4 = put_d(cp->h_alpha4, Nhol);
    t.s_alpha = put_d(cp->s_alpha, Nssh); t.s_alpha2 = put_d(cp->s_alpha2, Nssh); t.s_alpha3 = put_d(cp->s_alpha3, Nssh); t.s_alpha4 = put_d(cp->s_alpha4, Nssh);
    t.ph_sign = put_d(ph_sign.data(), ph_sign.size());
    if (g.is_cplx) {
        // T = ComplexF64: ssh_parameters.α::Vector{T} — the imaginary parts arrive in four more arrays of the struct (NULL = a real coupling)
        const std::vector<double> zero((size_t)Nssh, 0.0);
        t.s_alpha_im = put_d(cp->s_alpha_im ? cp->s_alpha_im : zero.data(), Nssh);
        t.s_alpha2_im = put_d(cp->s_alpha2_im ? cp->s_alpha2_im : zero.data(), Nssh);
        t.s_alpha3_im = put_d(cp->s_alpha3_im ? cp->s_alpha3_im : zero.data(), Nssh);
        t.s_alpha4_im = put_d(cp->s_alpha4_im ? cp->s_alpha4_im : zero.data(), Nssh);
    }
    if (!ib.empty()) HIPCHK(c, hipMemcpy(F.blob, ib.data(), ib.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!db.empty()) HIPCHK(c, hipMemcpy((char *)F.blob + int_bytes, db.data(), db.size() * sizeof(double), hipMemcpyHostToDevice));
    const size_t nx = (size_t)g.nw * g.Lt * std::max(Nph, 1);
    HIPCHK(c, hipMalloc(&F.d_x, nx * sizeof(double)));
    HIPCHK(c, hipMemset(F.d_x, 0, nx * sizeof(double)));
    // the force and the per-coupling contributions it is reduced from share one allocation: one memset per force evaluation clears both
    const size_t nx_pad = (nx + 1) & ~(size_t)1;  // keeps d_contrib 16-byte aligned
    HIPCHK(c, hipMalloc(&F.d_out, (nx_pad + (size_t)g.nw * g.Lt * std::max(Q, 1)) * sizeof(double)));
    F.d_contrib = F.d_out + nx_pad;
    HIPCHK(c, hipHostMalloc(&F.h_out, nx * sizeof(double)));
    HIPCHK(c, hipMalloc(&F.d_bare, ((size_t)g.N + 2 * (size_t)g.Nh + 1) * sizeof(double)));  // [V⁰ | Re t⁰ | Im t⁰ (complex T)] in checkerboard order
    F.Nph = Nph; F.Nhol = Nhol; F.Nssh = Nssh; F.Q = Q; F.dtau = cp->dtau; F.set = true;
    F.finite_mass.assign((size_t)std::max(Nph, 1), 1);
    for (int p = 0; p < Nph; ++p) F.finite_mass[p] = cp->finite_mass[p] ? 1 : 0;
    return 0;
}

int smoqy_force_set_phonons(smoqy_ctx *c, const double *x_all)
{
    CHECK_CTX(c);
    if (!c->force.set) FAIL(c, 1, "call smoqy_force_set_couplings first");
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (nx) HIPCHK(c, hipMemcpyAsync(c->force.d_x, x_all, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

ForceArgs force_args(smoqy_ctx *c, double nu, const double2 *u, const double2 *v)
{
    const Geometry &g = c->g;
    ForceArgs a = c->force.tmpl;
    a.Lt = g.Lt; a.N = g.N; a.Nh = g.Nh; a.ncol = g.ncol; a.nsys = g.nsys; a.nrhs = g.nrhs; a.nw = g.nw;
    a.Tc = 1; a.nchunk = g.Lt;  // one slice per workgroup: the kernel keeps two N-vectors per slice in LDS
    a.bonds = c->d_bonds; a.col_off = c->d_col_off; a.expV = c->d_expV; a.ch = c->d_ch; a.sh = c->d_sh; a.lam = c->d_lam;
    a.shi = c->d_shi;  // nullptr for real hoppings
    a.u = u; a.v = v; a.nu = nu; a.dtau = c->force.dtau;
    a.Nph = c->force.Nph; a.Nhol = c->force.Nhol; a.Nssh = c->force.Nssh; a.Q = c->force.Q;
    a.x = c->force.d_x; a.contrib = c->force.d_contrib;
    a.scratch = c->d_big; a.scratch_stride = c->big_stride;
    a.cs_slice0 = (c->g.is_sym && !c->cs_const.empty()) ? 1 : 0;
    for (int w = 0; w < g.nw && a.cs_slice0; ++w) a.cs_slice0 = c->cs_const[(size_t)w] != 0;
    return a;
}

// reduce the contribution slots into d_out (+=), bring it to the host and add it to `out`
static int force_finish(smoqy_ctx *c, const ForceArgs &a, double *out, bool fetch)
{
    launch_force_reduce(c->stream, a, c->force.d_out);
    if (!fetch) return check_launch(c, "force");
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    HIPCHK(c, hipMemcpyAsync(c->force.h_out, c->force.d_out, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = check_launch(c, "force")) return rc;
    for (size_t k = 0; k < nx; ++k) out[k] += c->force.h_out[k];
    return 0;
}

static int force_begin(smoqy_ctx *c)
{
    if (!c->force.set) FAIL(c, 1, "call smoqy_force_set_couplings first");
    const size_t nx = (size_t)c->g.nw * c->g.Lt * std::max(c->force.Nph, 1);
    HIPCHK(c, hipMemsetAsync(c->force.d_out, 0, (((nx + 1) & ~(size_t)1) + (size_t)c->g.nw * c->g.Lt * std::max(c->force.Q, 1)) * sizeof(double), c->stream));  // d_out and d_contrib (one allocation)
    return 0;
}

int smoqy_force_dMdx_v(smoqy_ctx *c, double nu, int u, int v, double *out)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, u)) return rc;
    if (int rc = check_vec(c, v)) return rc;
    if (int rc = force_begin(c)) return rc;
    ForceArgs a = force_args(c, nu, c->vecs[u], c->vecs[v]);
    if (!c->d_big && sizeof(double2) * 2 * (size_t)a.N > 160 * 1024 - 256) FAIL(c, 5, "N = %d does not fit the force kernel's LDS tile", a.N);
    launch_dmdx(c->stream, a, c->g.is_sym != 0, &c->ff);
    return force_finish(c, a, out, true);
}

int smoqy_force_dLdx_v(smoqy_ctx *c, double nu, int up, int u, double *out)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, up)) return rc;
    if (int rc = check_vec(c, u)) return rc;
    if (int rc = force_begin(c)) return rc;
    ForceArgs a = force_args(c, nu, c->vecs[up], c->vecs[u]);
    launch_dldx(c->stream, a);
    return force_finish(c, a, out, true);
}

// ΛΨ, AΨ = MΛΨ, ∂M/∂x term, MᵀAΨ, ∂Λ/∂x term; leaves the force of every walker in force.d_out
static int force_device(smoqy_ctx *c, int psi)
{
    if (int rc = check_vec(c, psi)) return rc;
    if (int rc = force_begin(c)) return rc;
    const Geometry &g = c->g;
    double2 *Psi = c->vecs[psi], *LPsi = c->scr[1], *APsi = c->scr[2], *MtAPsi = c->scr[0];
    launch_lambda_apply(c->stream, SMOQY_LAMBDA_MUL, LPsi, Psi, c->d_lam, g.Lt, g.N, g.nsys, g.nrhs, -1);           // ΛΨ            PFFCalculator.jl:146
    if (int rc = matvec_dev(c, SMOQY_OP_M, APsi, LPsi, nullptr, nullptr, 0, g.nsys)) return rc;                      // AΨ = MΛΨ      :148
    ForceArgs a = force_args(c, -2.0, APsi, LPsi);
    if (!c->d_big && sizeof(double2) * 2 * (size_t)a.N > 160 * 1024 - 256) FAIL(c, 5, "N = %d does not fit the force kernel's LDS tile", a.N);
    launch_dmdx(c->stream, a, g.is_sym != 0, &c->ff);                                                                      // -2 Re<AΨ|∂M/∂x|ΛΨ>   :150
    if (int rc = matvec_dev(c, SMOQY_OP_MT, MtAPsi, APsi, nullptr, nullptr, 0, g.nsys)) return rc;                   // MᵀAΨ          :153
    ForceArgs b = force_args(c, -2.0, MtAPsi, Psi);
    launch_dldx(c->stream, b);                                                                                       // -2 Re<MᵀAΨ|∂Λ/∂x|Ψ>  :155
    return force_finish(c, b, nullptr, false);
}

int smoqy_force_v(smoqy_ctx *c, int psi, double *out)
{
    CHECK_CTX(c);
    if (int rc = force_device(c, psi)) return rc;
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    HIPCHK(c, hipMemcpyAsync(c->force.h_out, c->force.d_out, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = check_launch(c, "force")) return rc;
    for (size_t k = 0; k < nx; ++k) out[k] += c->force.h_out[k];
    return 0;
}

int smoqy_set_bare_model(smoqy_ctx *c, const double *V0, const double *t0, const int64_t *perm)
{
    CHECK_CTX(c);
    if (!c->force.set) FAIL(c, 1, "call smoqy_force_set_couplings first");
    if (!V0 || (c->g.Nh && (!t0 || !perm))) FAIL(c, 1, "V0, t0 and perm must be given");
    const Geometry &g = c->g;
    std::vector<double> b((size_t)g.N + 2 * (size_t)g.Nh, 0.0);
    for (int i = 0; i < g.N; ++i) b[i] = V0[i];
    for (int n = 0; n < g.Nh; ++n) {
        if (perm[n] < 1 || perm[n] > g.Nh) FAIL(c, 1, "perm[%d] = %lld out of range", n + 1, (long long)perm[n]);
        // FermionDetMatrix.jl:224-228: sorted bond n is model hopping perm[n].  T = ComplexF64: t0 is complex128 (interleaved re, im)
        if (g.is_cplx) {
            b[(size_t)g.N + n] = t0[2 * (perm[n] - 1)];
            b[(size_t)g.N + g.Nh + n] = t0[2 * (perm[n] - 1) + 1];
        } else {
            b[(size_t)g.N + n] = t0[perm[n] - 1];
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(c->force.d_bare, b.data(), b.size() * sizeof(double), hipMemcpyHostToDevice));
    c->force.t0_level = (g.is_cplx || g.Nh == 0) ? 1 : cs_level_of(c, b.data() + g.N, 1);  // bare hoppings equal on every bond of a colour?
    c->force.bare_set = true;
    c->force.t_done = false;
    return 0;
}

int smoqy_update_from_phonons_all(smoqy_ctx *c, const double *x_all)
{
    CHECK_CTX(c);
    auto &F = c->force;
    if (!F.set || !F.bare_set) FAIL(c, 1, "call smoqy_force_set_couplings and smoqy_set_bare_model first");
    const Geometry &g = c->g;
    const size_t nx = (size_t)g.nw * g.Lt * F.Nph;
    if (nx) {
        if (!x_all) FAIL(c, 1, "x_all is NULL");
        HIPCHK(c, hipMemcpyAsync(F.d_x, x_all, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    ForceArgs a = force_args(c, 0.0, nullptr, nullptr);
    const bool do_t = g.Nh > 0 && (F.Nssh > 0 || !F.t_done);  // hoppings that no phonon couples to are refreshed once
    launch_phonon_fields(c->stream, a, F.d_bare, F.d_bare + g.N, c->d_expV, c->d_ch, c->d_sh, c->d_lam, g.is_sym ? F.dtau / 2 : F.dtau, do_t, g.is_cplx ? F.d_bare + g.N + g.Nh : nullptr,
                         c->d_shi);
    if (do_t) {
        launch_pack_csf(c->stream, c->d_ch, c->d_sh, c->d_psrc, c->d_csf, c->d_cs_varies, g.nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal, c->d_csi ? c->d_shi : nullptr, c->d_csi);
        for (int w = 0; w < g.nw; ++w) set_cs_const(c, w, F.Nssh == 0 ? F.t0_level : 0);  // no SSH coupling: t is the bare per-bond hopping on every slice
    }
    F.t_done = true;
    HIPCHK(c, hipStreamSynchronize(c->stream));  // x_all is the caller's again
    return check_launch(c, "update_from_phonons");
}

int smoqy_force_store_v(smoqy_ctx *c, int psi, double *out)
{
    CHECK_CTX(c);
    if (!out) FAIL(c, 1, "out is NULL");
    if (int rc = force_device(c, psi)) return rc;
    const size_t nx = (size_t)c->g.nw * c->g.Lt * c->force.Nph;
    if (nx) HIPCHK(c, hipMemcpyAsync(out, c->force.d_out, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "force");
}



// ---- calculate_derivative_fermionic_action! in one call ---------------------------------------------------

// calculate_derivative_fermionic_action! (src/PFFCalculator.jl:119-157) on the fields as they stand: [update_preconditioner!], Ψ = Λ⁻ᵀΦ,
// Ψ = (MᵀM)⁻¹Ψ, Ψ = Λ⁻¹Ψ, S_f = Φ·Ψ (left in d_dot_out), and with want_force the force in force.d_out.  No phonon-field upload, no
// force download: the callers decide what crosses the boundary.
int pff_core(smoqy_ctx *c, int phi, int psi, const double *randvec_all, double tol, int maxiter, int use_precond, bool want_force, int *iters, double *eps,
                    const double *d_randvec_all, double2 *d_dot, int async_step)
{
    const Geometry &g = c->g;
    if ((randvec_all || d_randvec_all) && use_precond) if (int rc = precond_update_range(c, 0, g.nw, randvec_all, d_randvec_all)) return rc;  // FermionDetMatrix.jl:259
    // Ψ = Λ⁻ᵀΦ (PFFCalculator.jl:97), ldiv!(Ψ, fdm, Ψ) (:99), Ψ = Λ⁻¹Ψ (:107) and the partials of S_f = Φ·Ψ (:109) in the kernels of the solve:
    // cg_init reads Φ through Λ⁻ᵀ, cg_finish writes Λ⁻¹x into the scratch vector that then becomes Ψ (CgArgs::lam)
    if (int rc = cg_dev(c, c->vecs[psi], nullptr, true, tol, maxiter, use_precond, iters, eps, c->vecs[phi], c->scr[0], async_step)) return rc;
    std::swap(c->scr[0], c->vecs[psi]);
    launch_dot_final(c->stream, c->part_c, d_dot ? d_dot : c->d_dot_out, g.nsys, c->nchunk);  // d_dot: a trajectory keeps the S_f of every step on the device until its end
    if (want_force) if (int rc = force_device(c, psi)) return rc;                                                               // :146-155
    return 0;
}

int smoqy_pff_step_v(smoqy_ctx *c, int phi, int psi, const double *x_all, const double *randvec_all, double tol, int maxiter, int use_precond, double *Sf, int *iters, double *eps, double *dSdx)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, phi)) return rc;
    if (int rc = check_vec(c, psi)) return rc;
    if (phi == psi) FAIL(c, 1, "phi and psi must be different vectors");
    const Geometry &g = c->g;
    if (g.nrhs != 1) FAIL(c, 1, "smoqy_pff_step_v needs a handle with nrhs = 1");
    if (dSdx && !c->force.set) FAIL(c, 1, "call smoqy_force_set_couplings first");
    if (x_all) if (int rc = smoqy_update_from_phonons_all(c, x_all)) return rc;                          // EFAPFFHMCUpdater.jl:200-205
    if (int rc = pff_core(c, phi, psi, randvec_all, tol, maxiter, use_precond, dSdx != nullptr, iters, eps)) return rc;
    if (dSdx) {
        const size_t nx = (size_t)g.nw * g.Lt * c->force.Nph;
        if (nx) HIPCHK(c, hipMemcpyAsync(dSdx, c->force.d_out, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    double2 *hdot = reinterpret_cast<double2 *>(c->h_poll_dot);
    HIPCHK(c, hipMemcpyAsync(hdot, c->d_dot_out, (size_t)g.nsys * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (int rc = check_launch(c, "pff_step")) return rc;
    if (Sf)
        for (int w = 0; w < g.nw; ++w) Sf[w] = hdot[w].x;
    return 0;
}


}  // extern "C"
