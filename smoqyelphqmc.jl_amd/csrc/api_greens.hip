// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "greens": GreensEstimator on a follower handle.
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- GreensEstimator -----------------------------------------------------------

int smoqy_copy_fields(smoqy_ctx *dst, int dst_walker, smoqy_ctx *src, int src_walker)
{
    CHECK_CTX(dst);
    if (!src) FAIL(dst, 1, "source handle is NULL");
    CHECK_WALKER(dst, dst_walker);
    if (src_walker < 0 || src_walker >= src->g.nw) FAIL(dst, 1, "source walker %d out of range", src_walker);
    const Geometry &a = dst->g, &b = src->g;
    if (a.Lt != b.Lt || a.N != b.N || a.Nh != b.Nh || a.ncol != b.ncol || a.is_sym != b.is_sym || a.is_cplx != b.is_cplx || dst->kg.ptotal != src->kg.ptotal || dst->device != src->device)
        FAIL(dst, 1, "smoqy_copy_fields needs two handles of the same lattice, propagator form and device");
    HIPCHK(dst, hipStreamSynchronize(src->stream));  // the source's fields are final
    const size_t nV = (size_t)a.Lt * a.N, nT = (size_t)a.Lt * a.Nh, nP = (size_t)a.Lt * dst->kg.ptotal;
    HIPCHK(dst, hipMemcpyAsync(dst->d_expV + dst_walker * nV, src->d_expV + src_walker * nV, nV * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(dst, hipMemcpyAsync(dst->d_lam + dst_walker * nV, src->d_lam + src_walker * nV, nV * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    if (nT) {
        HIPCHK(dst, hipMemcpyAsync(dst->d_ch + dst_walker * nT, src->d_ch + src_walker * nT, nT * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
        HIPCHK(dst, hipMemcpyAsync(dst->d_sh + dst_walker * nT, src->d_sh + src_walker * nT, nT * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
        if (a.is_cplx) HIPCHK(dst, hipMemcpyAsync(dst->d_shi + dst_walker * nT, src->d_shi + src_walker * nT, nT * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    }
    if (nP) HIPCHK(dst, hipMemcpyAsync(dst->d_csf + dst_walker * nP, src->d_csf + src_walker * nP, nP * sizeof(double2), hipMemcpyDeviceToDevice, dst->stream));
    if (nP && dst->d_csi && src->d_csi) HIPCHK(dst, hipMemcpyAsync(dst->d_csi + dst_walker * nP, src->d_csi + src_walker * nP, nP * sizeof(double), hipMemcpyDeviceToDevice, dst->stream));
    HIPCHK(dst, hipMemcpyAsync(dst->d_cs_varies + dst_walker, src->d_cs_varies + src_walker, sizeof(int), hipMemcpyDeviceToDevice, dst->stream));
    set_cs_const(dst, dst_walker, src->cs_const.empty() ? 0 : (int)src->cs_const[(size_t)src_walker]);
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    return 0;
}

int smoqy_ge_config(smoqy_ctx *c, int n_orbitals, int D, const int64_t *Ldims)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    if (n_orbitals < 1 || D < 1 || !Ldims) FAIL(c, 1, "invalid unit cell / lattice description");
    if (D > 2) FAIL(c, 5, "GreensEstimator contractions need a (D+1)-dimensional transform; rocFFT plans stop at 3 dimensions (D = %d)", D);
    size_t Nc = 1;
    for (int d = 0; d < D; ++d) {
        if (Ldims[d] < 1) FAIL(c, 1, "L[%d] = %lld", d, (long long)Ldims[d]);
        Nc *= (size_t)Ldims[d];
    }
    if ((size_t)n_orbitals * Nc != (size_t)g.N) FAIL(c, 1, "n_orbitals * prod(L) = %zu does not match N = %d", (size_t)n_orbitals * Nc, g.N);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ge_release(c);
    auto &G = c->ge;
    G.n_orb = n_orbitals; G.D = D; G.Nc = (int)Nc; G.n2 = 2 * (size_t)g.Lt * Nc;
    G.Ld[0] = (int)Ldims[0]; G.Ld[1] = D > 1 ? (int)Ldims[1] : 1;
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    size_t len[3] = {2 * (size_t)g.Lt, 1, 1};  // τ fastest, then the lattice directions: the reference's (2Lτ, L...) column-major arrays (:91-92)
    for (int d = 0; d < D; ++d) len[1 + d] = (size_t)Ldims[d];
    FFTCHK(c, rocfft_plan_create(&G.fwd_sys, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, (size_t)D + 1, len, (size_t)g.nsys, nullptr));
    FFTCHK(c, rocfft_plan_create(&G.inv_sys, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, (size_t)D + 1, len, (size_t)g.nsys, nullptr));
    FFTCHK(c, rocfft_plan_create(&G.inv_w, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, (size_t)D + 1, len, (size_t)g.nw, nullptr));
    size_t wsz = 0;
    for (rocfft_plan p : {G.fwd_sys, G.inv_sys, G.inv_w}) {
        size_t w1 = 0;
        FFTCHK(c, rocfft_plan_get_work_buffer_size(p, &w1));
        wsz = std::max(wsz, w1);
    }
    FFTCHK(c, rocfft_execution_info_create(&G.info));
    if (wsz) {
        HIPCHK(c, hipMalloc(&G.work, wsz));
        FFTCHK(c, rocfft_execution_info_set_work_buffer(G.info, G.work, wsz));
    }
    FFTCHK(c, rocfft_execution_info_set_stream(G.info, c->stream));
    HIPCHK(c, hipMalloc(&G.A, (size_t)g.nsys * G.n2 * sizeof(double2)));
    HIPCHK(c, hipMalloc(&G.B, (size_t)g.nsys * G.n2 * sizeof(double2)));
    HIPCHK(c, hipMalloc(&G.P, (size_t)g.nw * G.n2 * sizeof(double2)));
    HIPCHK(c, hipMalloc(&G.out, (size_t)g.nw * Nc * ((size_t)g.Lt + 1) * sizeof(double2)));
    HIPCHK(c, hipMalloc(&G.bpart, (size_t)g.nw * 64 * sizeof(double2)));
    HIPCHK(c, hipMalloc(&G.bout, (size_t)g.nw * sizeof(double2)));
    for (int q = 0; q < 2; ++q) HIPCHK(c, hipMalloc(&G.tw[q], (size_t)g.Lt * Nc * sizeof(double2)));
    // four-point estimators: (Lτ, L...) periodic transforms (cfft!/cifft!, :95-98), batched over the pairs of random vectors
    G.n1 = (size_t)g.Lt * Nc;
    G.npairs = g.nrhs * (g.nrhs - 1) / 2;
    if (G.npairs > 0) {
        size_t plen[3] = {(size_t)g.Lt, 1, 1};
        for (int d = 0; d < D; ++d) plen[1 + d] = (size_t)Ldims[d];
        FFTCHK(c, rocfft_plan_create(&G.pfwd, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, (size_t)D + 1, plen, (size_t)G.npairs, nullptr));
        FFTCHK(c, rocfft_plan_create(&G.pinv, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, (size_t)D + 1, plen, (size_t)G.npairs, nullptr));
        FFTCHK(c, rocfft_plan_create(&G.pinv1, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, (size_t)D + 1, plen, 1, nullptr));
        size_t pw = 0;
        for (rocfft_plan p : {G.pfwd, G.pinv, G.pinv1}) {
            size_t w1 = 0;
            FFTCHK(c, rocfft_plan_get_work_buffer_size(p, &w1));
            pw = std::max(pw, w1);
        }
        FFTCHK(c, rocfft_execution_info_create(&G.pinfo));
        if (pw) {
            HIPCHK(c, hipMalloc(&G.pwork, pw));
            FFTCHK(c, rocfft_execution_info_set_work_buffer(G.pinfo, G.pwork, pw));
        }
        FFTCHK(c, rocfft_execution_info_set_stream(G.pinfo, c->stream));
        for (int q = 0; q < 4; ++q) HIPCHK(c, hipMalloc(&G.S[q], (size_t)g.nsys * G.n1 * sizeof(double2)));
        HIPCHK(c, hipMalloc(&G.X, (size_t)G.npairs * G.n1 * sizeof(double2)));
        HIPCHK(c, hipMalloc(&G.Y, (size_t)G.npairs * G.n1 * sizeof(double2)));
        std::vector<int2> pr;
        for (int n = 0; n + 1 < g.nrhs; ++n)
            for (int m = n + 1; m < g.nrhs; ++m) pr.push_back(make_int2(n, m));  // :285-286
        HIPCHK(c, hipMalloc(&G.pairs, pr.size() * sizeof(int2)));
        HIPCHK(c, hipMemcpy(G.pairs, pr.data(), pr.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
    G.set = true;
    return 0;
}

int smoqy_ge_measure_pairs(smoqy_ctx *c, int gr, int r, const smoqy_ge_slot *slots, const void *tD, int conj_tD, const void *t0, int conj_t0, void *out)
{
    CHECK_CTX(c);
    auto &G = c->ge;
    if (!G.set) FAIL(c, 1, "call smoqy_ge_config first");
    if (G.npairs < 1) FAIL(c, 1, "the pair estimators need nrhs >= 2 random vectors");
    if (int rc = check_vec(c, gr)) return rc;
    if (int rc = check_vec(c, r)) return rc;
    if (!slots || !out) FAIL(c, 1, "slots / out is NULL");
    const Geometry &g = c->g;
    int second = 0;
    for (int q = 0; q < 4; ++q) {
        const smoqy_ge_slot &s = slots[q];
        if (s.source < 0 || s.source > 1 || s.orbital < 1 || s.orbital > G.n_orb) FAIL(c, 1, "slot %d: source %d / orbital %d invalid", q, s.source, s.orbital);
        if (s.second) second |= 1 << q;
        launch_ge_slot_gather(c->stream, c->vecs[s.source ? r : gr], G.S[q], g.Lt, g.N, g.nsys, G.n_orb, s.orbital - 1, G.Nc, G.Ld[0], G.Ld[1], (int)(s.shift[0] % G.Ld[0]),
                              G.D > 1 ? (int)(s.shift[1] % G.Ld[1]) : 0, s.source);  // Rt = conj(R)
    }
    if (tD) HIPCHK(c, hipMemcpyAsync(G.tw[0], tD, G.n1 * sizeof(double2), hipMemcpyHostToDevice, c->stream));
    if (t0) HIPCHK(c, hipMemcpyAsync(G.tw[1], t0, G.n1 * sizeof(double2), hipMemcpyHostToDevice, c->stream));
    FFTCHK(c, rocfft_execution_info_set_stream(G.pinfo, c->stream));
    const size_t nout = (size_t)G.Nc * ((size_t)g.Lt + 1);
    const double scale = 1.0 / ((double)G.n1 * (double)G.n1 * (double)G.npairs);  // two normalised inverse transforms, 1/Npairs (:306)
    for (int w = 0; w < g.nw; ++w) {
        const size_t off = (size_t)w * g.nrhs * G.n1;
        launch_ge_pair_product(c->stream, G.S[0] + off, G.S[1] + off, G.S[2] + off, G.S[3] + off, G.X, G.Y, G.pairs, G.npairs, G.n1, second, tD ? G.tw[0] : nullptr, conj_tD,
                               t0 ? G.tw[1] : nullptr, conj_t0);                              // :626-646
        void *bx[1] = {G.X}, *by[1] = {G.Y}, *bp[1] = {G.P};
        FFTCHK(c, rocfft_execute(G.pfwd, bx, nullptr, G.pinfo));                              // :686
        FFTCHK(c, rocfft_execute(G.pinv, by, nullptr, G.pinfo));                              // :687
        launch_ge_pair_reduce(c->stream, G.X, G.Y, G.P, G.npairs, G.n1);                      // :692, summed over the pairs
        FFTCHK(c, rocfft_execute(G.pinv1, bp, nullptr, G.pinfo));                             // :695
        launch_ge_finalize_pairs(c->stream, G.P, G.out + (size_t)w * nout, g.Lt, G.Nc, scale);  // :697-705
    }
    HIPCHK(c, hipMemcpyAsync(out, G.out, (size_t)g.nw * nout * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "ge_measure_pairs");
}

int smoqy_ge_measure_GD0(smoqy_ctx *c, int gr, int r, int a, int b, void *out)
{
    CHECK_CTX(c);
    auto &G = c->ge;
    if (!G.set) FAIL(c, 1, "call smoqy_ge_config first");
    if (int rc = check_vec(c, gr)) return rc;
    if (int rc = check_vec(c, r)) return rc;
    if (a < 1 || a > G.n_orb || b < 1 || b > G.n_orb) FAIL(c, 1, "orbitals (%d, %d) out of range 1..%d", a, b, G.n_orb);
    if (!out) FAIL(c, 1, "out is NULL");
    const Geometry &g = c->g;
    FFTCHK(c, rocfft_execution_info_set_stream(G.info, c->stream));
    launch_ge_gather(c->stream, c->vecs[gr], G.A, g.Lt, g.N, g.nsys, G.n_orb, a - 1, G.Nc, 0);   // _aperiodic_copyto!(A, GR_a_i)   :213
    launch_ge_gather(c->stream, c->vecs[r], G.B, g.Lt, g.N, g.nsys, G.n_orb, b - 1, G.Nc, 1);    // _aperiodic_copyto!(B, Rt_b_i), Rt = conj(R)  :214, :171
    void *bufA[1] = {G.A}, *bufB[1] = {G.B}, *bufP[1] = {G.P};
    FFTCHK(c, rocfft_execute(G.fwd_sys, bufA, nullptr, G.info));                                  // mul!(a, pfft!, a)   :686
    FFTCHK(c, rocfft_execute(G.inv_sys, bufB, nullptr, G.info));                                  // mul!(b, pifft!, b)  :687 (1/n folded into the scale below)
    launch_ge_product(c->stream, G.A, G.B, G.P, G.n2, g.nrhs, g.nw);                              // a .* b, summed over the random vectors  :692, :205-217
    FFTCHK(c, rocfft_execute(G.inv_w, bufP, nullptr, G.info));                                    // mul!(a, pifft!, a)  :695
    const double scale = 1.0 / ((double)G.n2 * (double)G.n2 * (double)g.nrhs);
    launch_ge_finalize_gd0(c->stream, G.P, G.out, g.Lt, G.Nc, g.nw, scale, a == b);               // :697-705, :219-227
    HIPCHK(c, hipMemcpyAsync(out, G.out, (size_t)g.nw * G.Nc * ((size_t)g.Lt + 1) * sizeof(double2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "ge_measure_GD0");
}

int smoqy_ge_boundary_dot(smoqy_ctx *c, int gr, int r, int orbital_gr, int orbital_r, const int64_t *shift, const void *tD, int conj_tD, const int64_t *tshift, const void *t0, int conj_t0, void *out)
{
    CHECK_CTX(c);
    auto &G = c->ge;
    if (!G.set) FAIL(c, 1, "call smoqy_ge_config first");
    if (int rc = check_vec(c, gr)) return rc;
    if (int rc = check_vec(c, r)) return rc;
    if (orbital_gr < 1 || orbital_gr > G.n_orb || orbital_r < 1 || orbital_r > G.n_orb) FAIL(c, 1, "orbitals (%d, %d) out of range 1..%d", orbital_gr, orbital_r, G.n_orb);
    if (!shift || !out) FAIL(c, 1, "shift / out is NULL");
    if ((tD == nullptr) != (t0 == nullptr)) FAIL(c, 1, "tD and t0 must be given together");
    if (tD && !tshift) FAIL(c, 1, "tshift is NULL");
    const Geometry &g = c->g;
    const size_t n1 = (size_t)g.Lt * G.Nc;
    if (tD) {
        HIPCHK(c, hipMemcpyAsync(G.tw[0], tD, n1 * sizeof(double2), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(G.tw[1], t0, n1 * sizeof(double2), hipMemcpyHostToDevice, c->stream));
    }
    const double scale = 1.0 / ((double)g.nrhs * (double)n1);  // 1 / (Nrv · length), :327
    launch_ge_boundary(c->stream, c->vecs[gr], c->vecs[r], G.bpart, G.bout, g.Lt, g.N, g.nsys, g.nrhs, G.n_orb, orbital_gr - 1, orbital_r - 1, G.Nc, G.Ld[0], G.Ld[1], (int)(shift[0] % G.Ld[0]),
                       G.D > 1 ? (int)(shift[1] % G.Ld[1]) : 0, tD ? G.tw[0] : nullptr, conj_tD, tD ? (int)(tshift[0] % G.Ld[0]) : 0, (tD && G.D > 1) ? (int)(tshift[1] % G.Ld[1]) : 0,
                       tD ? G.tw[1] : nullptr, conj_t0, 64, scale);
    if (int rc = pin_d2h(c, out, G.bout, (size_t)g.nw * sizeof(double2))) return rc;
    return check_launch(c, "ge_boundary_dot");
}


}  // extern "C"
