// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "operator": FermionDetMatrix applies (kernel selection), Λ applies, FourierTransformer.
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

extern "C" {

// ---- matvec -------------------------------------------------------------------------------------

// Run length (output slices per workgroup) of the streaming MᵀM kernel for a launch over `count` systems, 0 = use the chunked kernels.
// SMOQY_FDM_STREAM=R forces R (0 switches the kernel off) for A/B measurements.
static int stream_run_length(const smoqy_ctx *c, int count, bool cs_const)
{
    static const int env = tuning_env(kTuneFdmStream);
    const Geometry &g = c->g;
    if (!g.is_sym || g.is_cplx || !c->ff.enabled || c->d_big) return 0;
    // τ-dependent hoppings on three or more colours (optical SSH): measured even with the chunked kernel (11.3 against 11.9 µs at 16 systems,
    // 28.4 against 29.4 at 64, worse at the run lengths in between) — the two field sets of four colours cost the occupancy the pipeline
    // gains; those handles keep the chunked kernel unless a run length is forced
    static const bool forced = env >= 0;
    if (!cs_const && g.ncol >= 3 && !(forced || c->stream_R > 0)) return 0;
    if (!cs_const && g.ncol >= 3 && c->ff.threads > 256) return 0;  // (and they need the 256-lane instantiation: no 128-VGPR cap)
    int R = env;
    if (R < 0) R = c->stream_R;
    if (R < 0) {
        // automatic (measured on MI355X): from 16 systems per launch the streaming kernel wins (16 systems 17.0 -> 14.1 µs,
        // 128 systems 116.7 -> 75.4 µs); at 8 and fewer the owner-computes kernel does.  Run length: about 512 workgroups of 256 lanes per
        // launch, between 2 and 32 slices (small lattices want the short runs: honeycomb L = 8 at 16 systems 6.0 µs at R = 2, 7.6 at R = 4,
        // 7.8 chunked).
        if (count < 16) return 0;
        const long want = (long)g.Lt * count * c->ff.threads / (512L * 256L);
        R = 2;
        while (2 * R <= want && R < 32) R *= 2;
    }
    if (R <= 0) return 0;
    R = std::min(R, g.Lt);
    R -= R % c->Tc;
    return R >= 2 ? R : 0;
}

// Run length (output slices per WAVEFRONT) of fdm_wave_kernel for a launch over `count` systems.  A run of R slices costs 2R + 1 propagates
// and R + 2 slice loads, so long runs waste less; short runs give more wavefronts.  About two wavefronts per SIMD (2048 per launch) are
// wanted; the τ-chunk is the unit (the p·Ap partials keep the chunk layout).  SMOQY_FDM_WAVE_R forces a value.
static int wave_run_length(const smoqy_ctx *c, int count)
{
    static const int env = tuning_env(kTuneFdmWaveR);
    const Geometry &g = c->g;
    if (c->d_big || g.is_cplx || !g.is_sym) return 0;
    int R = env > 0 ? env : c->wave_R;
    if (R <= 0) {
        // automatic, from the measurements of tools/wave_scan.py on MI355X (µs per launch, wave kernel against the workgroup kernels):
        //   plaquette (optical-SSH square L = 12): 9.1 against 12.1 at 16 systems, 29.8 against 49.3 at 128 — always;
        //   honeycomb blocks (L = 16): 15.1 / 15.1 at 16 systems, 33.3 / 38.0 at 64 (R = 8), 75.0 / 80.0 at 128 (R = 16) — from 64 systems;
        //     L = 8 (a quarter of the lanes busy): 9.1 against 5.8 — never below 64 lanes... so only full wavefronts;
        //   ring (bond-SSH chain L = 256): 11.3 / 11.3 at 16, 81.2 / 82.1 at 128 — no gain: the workgroup kernels stay.
        // (at 16 systems every form moves ~4 TB/s out of the Infinity Cache with two slices of halo per run; longer runs have too few
        // wavefronts to hide one wavefront's chain of 2R + 1 dependent propagates)
        if (c->fw.kind == 1) return 0;
        if (c->fw.kind == 3 && (count < 64 || c->fw.lanes < 64)) return 0;
        // wavefronts a launch should keep: one per SIMD (1024) — which is all the honeycomb-block program's 264 registers admit, and what the
        // other programs were tuned at; two per SIMD for the honeycomb twin of SMOQY_FDM_WAVE_OCC=2 (kernels_fdm_wave.hip, not yet timed)
        static const int occ = tuning_env(kTuneFdmWaveOcc);
        const long want = (c->fw.kind == 3 && occ == 2) ? 2048 : 1024;
        R = c->Tc;
        while (2 * R <= 32 && (long)((g.Lt + 2 * R - 1) / (2 * R)) * count >= want) R *= 2;
    }
    R = std::min(R, g.Lt);
    R -= R % c->Tc;
    return std::max(R, 0);
}

int matvec_dev(smoqy_ctx *c, int op, double2 *out, const double2 *in, double2 *partial, const CgState *cg, int sys0, int count, bool twiddled, hipStream_t st)
{
    if (!st) st = c->stream;
    if (op < SMOQY_OP_M || op > SMOQY_OP_MMT) FAIL(c, 1, "unknown matvec op %d", op);
    FdmArgs a = fdm_args(c, in, out, partial, cg, sys0, count);
    if (twiddled) {  // Θ M Θᴴ: uniform hop phase exp(-iπ/Lτ), periodic in τ (kernels_vec.hip, CG section)
        a.hop_re = std::cos(M_PI / c->g.Lt);
        a.hop_im = -std::sin(M_PI / c->g.Lt);
        a.antiperiodic = 0;
    }
    auto &T = c->mvt;
    const bool sample = T.every > 0 && op == SMOQY_OP_MTM && count == c->g.nsys && T.used < (int)T.ev.size() && (T.seen++ % T.every) == 0;
    if (sample) {
        HIPCHK(c, hipEventRecord(T.ev[T.used].first, c->stream));
        if (T.d_stamp && T.used < T.stamp_cap && a.nchunk * a.sys_count <= T.stamp_wgs) a.stamp = T.d_stamp + 2 * (size_t)T.stamp_wgs * T.used;  // register-resident kernels only
    }
    bool cs_const = c->g.is_sym != 0 && !c->cs_const.empty();
    for (int w = sys0 / c->g.nrhs; cs_const && w <= (sys0 + count - 1) / c->g.nrhs; ++w) cs_const = c->cs_const[(size_t)w] != 0;
    // streaming MᵀM (fdm_stream_kernel): workgroups walk runs of slices with their loads two iterations ahead — for launches big enough that
    // the chunked kernel's load-wait-compute workgroups leave the memory system idle
    a.run_len = (op == SMOQY_OP_MTM && in != out) ? stream_run_length(c, count, cs_const) : 0;
    {   // exp(-ΔτV) is read once per launch: where the vectors of several solves compete for the Infinity Cache (the in-place τ-FFT form is
        // the sign of it, see cg_iteration_fused) it is loaded past the caches (+1.1 % in the eight-stream bench, four alternating pairs)
        static const int nt_env = tuning_env(kTuneNtFields);  // A/B switch
        a.nt_fields = nt_env < 0 ? (c->tf_ok && c->tf.slim) : (nt_env != 0);
    }
    const char *name;
    // one wavefront per run of slices, the slice in registers (kernels_fdm_wave.hip): lattices with a lane program, fused MᵀM out of place
    int csm = 2;  // what the host has shown for EVERY walker of the launch: 2 τ-dependent (or unknown), 1 τ-independent, 0 and uniform per colour
    if (cs_const) {
        csm = 0;
        for (int w = sys0 / c->g.nrhs; w <= (sys0 + count - 1) / c->g.nrhs; ++w) csm = std::max(csm, c->cs_const[(size_t)w] >= 2 ? 0 : 1);
    }
    const int wave_R = (op == SMOQY_OP_MTM && in != out && c->fw.kind && !c->wave_off) ? wave_run_length(c, count) : 0;
    FdmArgs aw = a;
    aw.run_len = wave_R;
    if (wave_R > 0 && fdm_wave_supported(aw, c->ff, c->fw, c->g.is_sym != 0, csm)) {
        launch_fdm_wave(st, aw, c->ff, c->fw, csm);
        name = c->fw.kind == 1 ? "fdm_wave_kernel<ring>" : (c->fw.kind == 2 ? "fdm_wave_kernel<plaquette>" : (tuning_env(kTuneFdmWaveOcc) == 2 ? "fdm_wave_kernel<honeycomb block, two wavefronts per SIMD>" : "fdm_wave_kernel<honeycomb block>"));
    } else if (a.run_len > 0 && fdm_own_stream_supported(a, c->ff, c->g.is_sym != 0, cs_const)) { launch_fdm_own_stream(st, a, c->ff); name = "fdm_own_stream_kernel"; }
    else if (a.run_len > 0 && fdm_stream_supported(a, c->ff, c->g.is_sym != 0)) { launch_fdm_stream(st, a, c->ff, cs_const); name = cs_const ? "fdm_stream_kernel<CSV=false>" : "fdm_stream_kernel<CSV=true>"; }
    else if (fdm_own_supported(a, c->ff, c->g.is_sym != 0)) { launch_fdm_own(st, op, a, c->ff); name = "fdm_own_kernel"; }
    else if (fdm_fast_supported(a, c->ff, c->g.is_sym != 0)) { launch_fdm_fast(st, op, a, c->ff, c->g.is_sym != 0, cs_const); name = c->g.is_sym ? "fdm_fast_kernel" : "fdm_fast_asym_kernel"; }
    else { launch_fdm(st, op, c->g.is_sym != 0, a, c->d_big ? 0 : fdm_lds_bytes(op, c->g.N, c->Tc)); name = "fdm_kernel"; }
    if (op == SMOQY_OP_MTM && count == c->g.nsys) c->mtm_name = name;
    if (sample) HIPCHK(c, hipEventRecord(T.ev[T.used++].second, c->stream));
    return check_launch(c, "matvec");
}

// run length of the streaming MᵀM kernel: -1 = automatic (the default), 0 = chunked kernels only, R >= 2 = workgroups walk runs of R
// slices (rounded down to a multiple of the τ-chunk)
// run length of the one-wavefront-per-run MᵀM kernel: -1 = automatic (the default), 0 = never use it for this handle, R >= 1 = runs of R slices
int smoqy_matvec_wave(smoqy_ctx *c, int run_len)
{
    CHECK_CTX(c);
    if (run_len < -1) FAIL(c, 1, "run_len must be -1 (automatic), 0 (off) or >= 1");
    c->wave_R = run_len;
    c->wave_off = run_len == 0;
    drop_graphs(c);
    return 0;
}

int smoqy_matvec_stream(smoqy_ctx *c, int run_len)
{
    CHECK_CTX(c);
    if (run_len < -1) FAIL(c, 1, "run_len must be -1 (automatic), 0 (off) or >= 2");
    c->stream_R = run_len;
    drop_graphs(c);
    return 0;
}

int smoqy_matvec_v(smoqy_ctx *c, int op, int out, int in)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, out)) return rc;
    if (int rc = check_vec(c, in)) return rc;
    if (out == in) {  // lmul_M!/lmul_Mt! (:372, :470): result lands in a scratch buffer that then becomes the vector
        if (int rc = matvec_dev(c, op, c->scr[0], c->vecs[in], nullptr, nullptr, 0, c->g.nsys)) return rc;
        std::swap(c->scr[0], c->vecs[out]);
        return 0;
    }
    return matvec_dev(c, op, c->vecs[out], c->vecs[in], nullptr, nullptr, 0, c->g.nsys);
}

int smoqy_matvec(smoqy_ctx *c, int op, void *out, const void *in, int sys0, int count)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_into(c, c->scr[1], in, sys0, count)) return rc;
    if (int rc = matvec_dev(c, op, c->scr[2], c->scr[1], nullptr, nullptr, sys0, count)) return rc;
    return download_from(c, c->scr[2], out, sys0, count);
}

// checkerboard_lmul! / checkerboard_ldiv! on a colour interval, in place (src/checkerboard_matrix_multiply.jl:26-145)
int smoqy_checkerboard_v(smoqy_ctx *c, int id, int inverse, int transposed, int color_first, int ncolors)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    if (color_first < 0 || ncolors < 0 || color_first + ncolors > c->g.ncol) FAIL(c, 1, "colour interval [%d, %d) outside 0..%d", color_first, color_first + ncolors, c->g.ncol);
    FdmArgs a = fdm_args(c, c->vecs[id], c->vecs[id], nullptr, nullptr, 0, c->g.nsys);
    launch_checkerboard(c->stream, a, inverse, transposed, color_first, ncolors);
    return check_launch(c, "checkerboard");
}

int smoqy_checkerboard(smoqy_ctx *c, void *inout, int inverse, int transposed, int color_first, int ncolors, int sys0, int count)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    if (color_first < 0 || ncolors < 0 || color_first + ncolors > c->g.ncol) FAIL(c, 1, "colour interval [%d, %d) outside 0..%d", color_first, color_first + ncolors, c->g.ncol);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_into(c, c->scr[1], inout, sys0, count)) return rc;
    FdmArgs a = fdm_args(c, c->scr[1], c->scr[1], nullptr, nullptr, sys0, count);
    launch_checkerboard(c->stream, a, inverse, transposed, color_first, ncolors);
    if (int rc = check_launch(c, "checkerboard")) return rc;
    return download_from(c, c->scr[1], inout, sys0, count);
}

// ---- Λ ------------------------------------------------------------------------------------------

int smoqy_lambda_set(smoqy_ctx *c, int w, const double *Lambda)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_real_field(c, Lambda, c->d_lam + (size_t)w * c->g.Lt * c->g.N, c->g.N)) return rc;
    return check_launch(c, "lambda_set");
}

int smoqy_lambda_get(smoqy_ctx *c, int w, double *Lambda)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    return download_real_field(c, c->d_lam + (size_t)w * c->g.Lt * c->g.N, Lambda, c->g.N);
}

static int lambda_update_range(smoqy_ctx *c, int w0, int nw, const double *x, int Nph, double dtau, int ncoup, const int64_t *c2p, const int64_t *c2s, const double *alpha, const double *alpha3, const int32_t *ph_sym)
{
    const Geometry &g = c->g;
    if (Nph < 0 || ncoup < 0) FAIL(c, 1, "negative Nph/ncoup");
    const size_t nx = (size_t)nw * Nph * g.Lt;
    if (int rc = ensure_stage_real(c, nx + 2 * (size_t)ncoup + 1)) return rc;
    if (int rc = ensure_stage_int(c, 4 * (size_t)ncoup + g.N + 1)) return rc;
    // [c2p | c2s | ph_sym | site_next(ncoup) | site_first(N)]; per-site coupling lists keep the reference's coupling order
    std::vector<int> ib(4 * (size_t)ncoup + g.N, -1);
    std::vector<int> last((size_t)g.N, -1);
    for (int k = 0; k < ncoup; ++k) {
        if (c2p[k] < 1 || c2p[k] > Nph || c2s[k] < 1 || c2s[k] > g.N) FAIL(c, 1, "coupling %d maps to phonon %lld / site %lld out of range", k + 1, (long long)c2p[k], (long long)c2s[k]);
        ib[k] = (int)c2p[k] - 1;
        ib[ncoup + k] = (int)c2s[k] - 1;
        ib[2 * ncoup + k] = ph_sym[k] ? 1 : 0;
        const int site = (int)c2s[k] - 1;
        if (last[site] < 0) ib[4 * (size_t)ncoup + site] = k;
        else ib[3 * (size_t)ncoup + last[site]] = k;
        last[site] = k;
    }
    if (nx) HIPCHK(c, hipMemcpyAsync(c->d_stage_real, x, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // the force kernels read the same phonon fields: keep their device copy in step (no second transfer)
    if (nx && c->force.set && c->force.Nph == Nph)
        HIPCHK(c, hipMemcpyAsync(c->force.d_x + (size_t)w0 * g.Lt * Nph, c->d_stage_real, nx * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (ncoup) {
        if (int rc = pin_h2d(c, c->d_stage_real + nx, alpha, (size_t)ncoup * sizeof(double))) return rc;
        if (int rc = pin_h2d(c, c->d_stage_real + nx + ncoup, alpha3, (size_t)ncoup * sizeof(double))) return rc;
    }
    if (int rc = pin_h2d(c, c->d_stage_int, ib.data(), ib.size() * sizeof(int))) return rc;
    launch_lambda_update(c->stream, c->d_lam + (size_t)w0 * g.Lt * g.N, nw * g.Lt, g.N, c->d_stage_real, Nph, dtau, ncoup, c->d_stage_int, c->d_stage_int + ncoup, c->d_stage_real + nx, c->d_stage_real + nx + ncoup,
                         c->d_stage_int + 2 * ncoup, c->d_stage_int + 4 * (size_t)ncoup, c->d_stage_int + 3 * (size_t)ncoup, g.Lt);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "lambda_update");
}

int smoqy_lambda_update(smoqy_ctx *c, int w, const double *x, int Nph, double dtau, int ncoup, const int64_t *c2p, const int64_t *c2s, const double *alpha, const double *alpha3, const int32_t *ph_sym)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    return lambda_update_range(c, w, 1, x, Nph, dtau, ncoup, c2p, c2s, alpha, alpha3, ph_sym);
}

int smoqy_lambda_update_all(smoqy_ctx *c, const double *x_all, int Nph, double dtau, int ncoup, const int64_t *c2p, const int64_t *c2s, const double *alpha, const double *alpha3, const int32_t *ph_sym)
{
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    return lambda_update_range(c, 0, c->g.nw, x_all, Nph, dtau, ncoup, c2p, c2s, alpha, alpha3, ph_sym);
}

int smoqy_lambda_apply_v(smoqy_ctx *c, int op, int out, int in)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, out)) return rc;
    if (int rc = check_vec(c, in)) return rc;
    if (op < 0 || op > 3) FAIL(c, 1, "unknown lambda op %d", op);
    const Geometry &g = c->g;
    if (out == in) {
        launch_lambda_apply(c->stream, op, c->scr[0], c->vecs[in], c->d_lam, g.Lt, g.N, g.nsys, g.nrhs, -1);
        std::swap(c->scr[0], c->vecs[out]);
    } else {
        launch_lambda_apply(c->stream, op, c->vecs[out], c->vecs[in], c->d_lam, g.Lt, g.N, g.nsys, g.nrhs, -1);
    }
    return check_launch(c, "lambda_apply");
}

int smoqy_lambda_apply(smoqy_ctx *c, int op, void *out, const void *in, const double *Lambda, int sys0, int count)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    if (op < 0 || op > 3) FAIL(c, 1, "unknown lambda op %d", op);
    const Geometry &g = c->g;
    const int w = sys0 / g.nrhs;
    if (Lambda) if (int rc = smoqy_lambda_set(c, w, Lambda)) return rc;
    if (int rc = upload_into(c, c->scr[1], in, sys0, count)) return rc;
    launch_lambda_apply(c->stream, op, c->scr[2], c->scr[1], c->d_lam, g.Lt, g.N, g.nsys, g.nrhs, Lambda ? w : -1);
    if (int rc = check_launch(c, "lambda_apply")) return rc;
    return download_from(c, c->scr[2], out, sys0, count);
}

// ---- FourierTransformer ---------------------------------------------------------------------------

static int fft_dev(smoqy_ctx *c, double2 *v, bool inverse)
{
    const Geometry &g = c->g;
    if (c->tf_ok && c->use_tfft) {  // twiddle fused into the transform's load / store
        TfftArgs t = c->tf;
        t.src = v; t.dst = v;
        t.pre_tw = inverse ? nullptr : c->d_tw;   // FourierTransformer.jl:46-47
        t.post_tw = inverse ? c->d_tw : nullptr;  // :60-61 (1/Lτ of the inverse = the two 1/√Lτ factors)
        launch_tfft(c->stream, inverse ? 1 : 0, t);
        return check_launch(c, "fft");
    }
    void *buf[1] = {v};
    if (!inverse) {
        launch_fft_twiddle(c->stream, v, c->d_tw, g.Lt, g.N, g.nsys, 0);  // FourierTransformer.jl:46
        FFTCHK(c, rocfft_execute(c->plan_f, buf, nullptr, c->fft_info)); // :47
    } else {
        FFTCHK(c, rocfft_execute(c->plan_b, buf, nullptr, c->fft_info)); // :60 (rocFFT's inverse carries no 1/n)
        launch_fft_twiddle(c->stream, v, c->d_tw, g.Lt, g.N, g.nsys, 1);  // :61 with the 1/Lτ folded in
    }
    return check_launch(c, "fft");
}

int smoqy_fft_use_rocfft(smoqy_ctx *c, int on)
{
    CHECK_CTX(c);
    c->use_tfft = on ? 0 : 1;
    drop_graphs(c);
    return 0;
}

int smoqy_fft_forward_v(smoqy_ctx *c, int id)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    return fft_dev(c, c->vecs[id], false);
}

int smoqy_fft_inverse_v(smoqy_ctx *c, int id)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    return fft_dev(c, c->vecs[id], true);
}

static int fft_host(smoqy_ctx *c, void *inout, int sys0, int count, bool inverse)
{
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_into(c, c->scr[1], inout, sys0, count)) return rc;
    if (int rc = fft_dev(c, c->scr[1], inverse)) return rc;
    return download_from(c, c->scr[1], inout, sys0, count);
}

int smoqy_fft_forward(smoqy_ctx *c, void *inout, int sys0, int count) { CHECK_CTX(c); return fft_host(c, inout, sys0, count, false); }
int smoqy_fft_inverse(smoqy_ctx *c, void *inout, int sys0, int count) { CHECK_CTX(c); return fft_host(c, inout, sys0, count, true); }


}  // extern "C"
