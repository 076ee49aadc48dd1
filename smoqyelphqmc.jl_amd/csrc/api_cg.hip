// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "cg": the on-device conjugate-gradient driver (bursts, polls, graphs, part streams, gate).
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

namespace {
// Process-wide gate on the CG loops (smoqy_cg_gate): with several handles driven by several host threads on one GPU, at most `limit`
// of them are inside a CG solve at once — the loops are bandwidth bound and a fourth concurrent one only evicts the others' working sets
// (measured) — while everything around the solve (preconditioner update, force, leapfrog, transfers) still overlaps freely.
struct CgGate {
    std::mutex m;
    std::condition_variable cv;
    int limit = 0, inside = 0;
    void enter()
    {
        std::unique_lock<std::mutex> lk(m);
        if (limit <= 0) { ++inside; return; }
        cv.wait(lk, [&] { return limit <= 0 || inside < limit; });
        ++inside;
    }
    void leave()
    {
        { std::lock_guard<std::mutex> lk(m); --inside; }
        cv.notify_one();
    }
} g_cg_gate;
struct CgGateHold {
    CgGateHold() { g_cg_gate.enter(); }
    ~CgGateHold() { g_cg_gate.leave(); }
};
}  // namespace

extern "C" {

// ---- conjugate gradient ---------------------------------------------------------------------------

int smoqy_cg_use_graph(smoqy_ctx *c, int on)
{
    CHECK_CTX(c);
    c->use_graph = on ? 1 : 0;
    return 0;
}

// state of the graph replay: *enabled = the switch as it stands (a failed capture clears it), *captured = live cached graphs;
// returns 0; smoqy_last_error holds the reason of the last failed capture
int smoqy_cg_graph_status(smoqy_ctx *c, int *enabled, int *captured)
{
    CHECK_CTX(c);
    int n = 0;
    for (auto &gph : c->graphs) n += (gph.exec && gph.epoch == c->graph_epoch) ? 1 : 0;
    if (enabled) *enabled = c->use_graph;
    if (captured) *captured = n;
    if (!c->graph_note.empty()) c->err = c->graph_note;
    return 0;
}

// process-wide: at most max_concurrent handles inside a CG solve at once (0 = no limit, the default); takes effect for solves that start later
int smoqy_cg_gate(int max_concurrent)
{
    { std::lock_guard<std::mutex> lk(g_cg_gate.m); g_cg_gate.limit = max_concurrent > 0 ? max_concurrent : 0; }
    g_cg_gate.cv.notify_all();
    return 0;
}

// multi-part pipeline of the CG loop: parts = 0 automatic (two parts from 8 systems up), 1 = off, 2..4 = that many parts
int smoqy_cg_split(smoqy_ctx *c, int parts)
{
    CHECK_CTX(c);
    if (parts < 0 || parts > smoqy_ctx::kMaxParts) FAIL(c, 1, "parts must be 0 (automatic) or 1..%d", smoqy_ctx::kMaxParts);
    c->cg_parts = parts;
    return set_part_streams(c, std::min(parts == 0 ? auto_parts(c) : parts, c->g.nsys));
}

// Where tfft_rb_kernel (Lτ = 80, 100, 200) runs, from the measurements in profiles/r04_rb_bench_ab.txt and r04_rb_rule_scan.txt:
//  * a handle left at the library's default form (one stream): up to 64 systems per launch — one-stream sweeps at 16 / 32 / 64 walkers
//    are 13-27 / 7-12 / 1-13 % shorter than with the two-image / in-place forms; larger launches were not measured and keep the in-place form;
//  * a handle whose caller asked for the in-place form (smoqy_tfft_form(1): several handles share the GPU, bench.py's timed batches):
//    below 32 systems per launch (chain, 8 x 16: +3.7 %); at 4 x 64 its 92-109 VGPRs (three or four 320-lane workgroups per CU against
//    five or six of 256) lose 3-8 %, at 8 x 32 the optical-SSH lattice gains 6.5 % and the chain loses 2 %.
// SMOQY_TFFT_EDGE=3 lifts every limit, =1 switches the kernel off (A/B aids).
void tfft_rb_rule(smoqy_ctx *c, bool shared_gpu)
{
    const int nsys = c->g.nsys;
    const bool on = tuning_env(kTuneTfftEdge) == 3 || nsys < 32 || (!shared_gpu && nsys <= 64);
    c->tf.rb = on ? c->tf_rb_plan : 0;
}

// form of the handle's own τ-FFT: 0 = two LDS images (fewer passes: fastest alone), 1 = in place (fewer registers and half the LDS: more
// workgroups per CU, better when several handles share the GPU)
int smoqy_tfft_form(smoqy_ctx *c, int in_place)
{
    CHECK_CTX(c);
    if (in_place != 0 && in_place != 1) FAIL(c, 1, "in_place must be 0 or 1");
    const int want = (in_place && c->tf_ok && c->tf.slim_ok && c->tf.pos) ? 1 : 0;  // lengths with a factor 7 keep the two-image form
    const int rb_before = c->tf.rb;
    tfft_rb_rule(c, in_place != 0);
    if (want != c->tf.slim || rb_before != c->tf.rb) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->tf.slim = want;
        drop_graphs(c);
    }
    return 0;
}

int smoqy_cg_config(smoqy_ctx *c, int check_every)
{
    CHECK_CTX(c);
    if (check_every < 1) FAIL(c, 1, "check_every must be >= 1");
    c->check_every = check_every;
    return 0;
}

// one CG iteration: ConjugateGradient.jl:216-246
// Part streams of the CG pipeline: exactly nparts - 1 of them exist.  They are created where the number of parts is DECIDED (smoqy_create
// for the automatic choice, smoqy_cg_split for an explicit one), not on first use, and surplus ones are destroyed: the runtime binds a new
// stream to the least-used of its few hardware queues, so a part stream created right after the handle's own stream lands on a different
// queue (one handle, eager: 16 walkers 183 sweeps/s; created lazily in mid-run: 167), while idle part streams of handles that do not split
// push the main streams of several handles onto the same queue (six handles with three idle streams each: 335 -> 251 sweeps/s).
int set_part_streams(smoqy_ctx *c, int nparts)
{
    for (int q = 0; q < smoqy_ctx::kMaxParts - 1; ++q) {
        const bool want = q + 1 < nparts;
        if (want && !c->part_stream[q]) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->part_stream[q], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_part[q], hipEventDisableTiming));
        } else if (!want && c->part_stream[q]) {
            HIPCHK(c, hipStreamSynchronize(c->part_stream[q]));
            (void)hipEventDestroy(c->ev_part[q]);
            (void)hipStreamDestroy(c->part_stream[q]);
            c->part_stream[q] = nullptr;
            c->ev_part[q] = nullptr;
        }
    }
    return 0;
}

int auto_parts(const smoqy_ctx *c) { return c->g.nsys >= 8 ? 2 : 1; }  // measured, DESIGN.md §4.3

// the fused form of one CG iteration for systems [sys0, sys0 + count) on stream st: four launches
static int cg_iteration_fused(smoqy_ctx *c, const CgArgs &a, hipStream_t st, int sys0, int count)
{
    // smoqy_cg_iteration_timing: full-batch iterations on the handle's own stream get an event in front of each launch and one behind the last
    auto &IT = c->itt;
    hipEvent_t *tev = (IT.used < IT.want && st == c->stream && count == c->g.nsys && !c->use_graph) ? &IT.ev[(size_t)5 * IT.used] : nullptr;
    if (tev) (void)hipEventRecord(tev[0], st);
    if (int rc = matvec_dev(c, SMOQY_OP_MTM, c->cg_z, c->cg_p, c->part_pz, c->d_st, sys0, count, true, st)) return rc;  // z = A p, partial p·Ap (:219)
    if (tev) (void)hipEventRecord(tev[1], st);
    // the tau-FFT kernels absorb the BLAS-1 updates (kernels_tfft.hip).  In this form a.r holds the residual in FREQUENCY space (r̂):
    // the forward kernel updates it with α·FFT(Ap), the Chebyshev kernel reads it and writes ẑ into v (out of place), the inverse
    // kernel turns ẑ into z and updates x and p.
    TfftArgs t = c->tf;
    t.sys_first = sys0; t.sys_count = count;
    t.x = a.x; t.r = a.r; t.p = a.p; t.z = a.z;
    // x is touched by the inverse kernel only, once per iteration.  Where the in-place form is in use — several handles share the GPU, or the
    // launch is HBM resident — the vectors of the solves in flight compete for the Infinity Cache, and x goes past it with nontemporal
    // loads and stores so that p, A p, r̂ and ẑ (each written by one kernel and read by the next) keep their hits: +2.4 % in the
    // eight-stream bench (four alternating pairs); a single small batch keeps x cached (one walker: 40.9 against 47.3 ms per sweep)
    static const int xs_env = tuning_env(kTuneXStream);  // A/B switch
    t.x_stream = xs_env < 0 ? t.slim : (xs_env != 0);
    // Workgroups go to the eight XCDs round-robin.  When a launch covers a multiple of eight systems the τ-FFT and Chebyshev kernels take the
    // blockIdx -> system map the MᵀM kernels already use (XCD x works on the contiguous share [x·n/8, (x+1)·n/8) of the systems), so that in
    // every kernel of the iteration an XCD touches the same eighth of each vector.  Measured: 8 walkers on one stream 57.7 -> 54.2 ms per
    // sweep, 16: 79.9 -> 79.3, bond-SSH chain 16 walkers 55.7 -> 46.5, 32 walkers of the headline lattice 135.1 -> 138.0 (hence the size
    // rule), the eight-stream bench unchanged.  It is NOT inter-kernel L2 reuse: FETCH_SIZE per kernel is the same with either map
    // (8 walkers: 16.5 / 16.3 MB for the forward τ-FFT) — the L2s do not keep lines across kernel boundaries; what shrinks is the address
    // range an XCD walks per kernel (translation and fabric locality).
    static const int xm_env = tuning_env(kTuneXcdMap);  // A/B switch
    const size_t xcd_share = (size_t)(count / 8) * 4 * c->g.Lt * c->g.N * sizeof(double2);
    t.xcd_map = xm_env < 0 ? (count % 8 == 0 && xcd_share <= (size_t)8 << 20) : (xm_env != 0);
    t.part_rz = a.part_rz; t.nrz = a.nrz; t.rz_stride = a.rz_stride;
    t.part_pz = a.part_pz; t.npz = a.nchunk; t.pz_stride = a.nchunk;
    t.part_rr = a.part_rr; t.nrr = t.ntile; t.rr_stride = c->pstride;
    t.st = a.st;
    launch_tfft(st, 2, t);                      // :219-226: α, r̂ -= α FFT(Ap), |r|²
    if (tev) (void)hipEventRecord(tev[2], st);
    KpmArgs k = kpm_args(c, c->cg_r, c->d_st);
    k.sys_first = sys0; k.sys_count = count;
    k.xcd_map = t.xcd_map;
    k.vout = c->cg_z;  // ẑ reuses the buffer of A p, which the forward kernel has consumed (one vector less in the cache-resident working set)
    k.part_rz = c->part_rz;
    launch_cheb(st, k, c->kg);                  // :237 in frequency space, partial r·z by Parseval
    c->cheb_name = cheb_kernel_name(k, c->kg);
    if (tev) (void)hipEventRecord(tev[3], st);
    t.src = a.v;
    launch_tfft(st, 3, t);                      // inverse FFT + x += α p + :229-245
    if (tev) { (void)hipEventRecord(tev[4], st); ++IT.used; }
    return check_launch(c, "cg iteration");
}

// one CG iteration: ConjugateGradient.jl:216-246
static int cg_iteration(smoqy_ctx *c, const CgArgs &a, bool any_pre)
{
    if (any_pre && c->tf_ok && c->use_tfft) return cg_iteration_fused(c, a, c->stream, 0, c->g.nsys);
    if (int rc = matvec_dev(c, SMOQY_OP_MTM, c->cg_z, c->cg_p, c->part_pz, c->d_st, 0, c->g.nsys, true)) return rc;  // z = A p, partial p·Ap (:219)
    launch_cg_update_xr(c->stream, a);                                                                             // :220-226
    if (any_pre) if (int rc = precond_core(c, c->cg_r, c->cg_z, c->d_st, c->part_rz)) return rc;                 // z = P⁻¹ r (over A p, consumed), partial r·z (:237-240)
    launch_cg_update_p(c->stream, a);                                                                              // :229-245
    return 0;
}

constexpr int kGraphIters = 4;  // iterations per captured graph (smoqy_cg_use_graph)

// pff_phi / pff_out: the solve of calculate_fermionic_action! with its Λ applies folded in (CgArgs::lam): b = Λ⁻ᵀ·pff_phi is formed by
// cg_init (b itself is not read), Ψ = Λ⁻¹x lands in pff_out with the partials of Φ·Ψ in part_c; x then holds the twiddled iterate only.
// async_step >= 0 (smoqy_hmc_trajectory_v only; needs the PFFCalculator form and x === b): launch c->traj_hint[async_step] + margin iterations,
// the finish kernel and a device-side copy of the states into slot async_step of c->d_traj_st, and return WITHOUT waiting; iters / eps are
// not touched, the caller verifies the states at the end of the trajectory.
int cg_dev(smoqy_ctx *c, double2 *x, const double2 *b, bool x_is_b, double tol, int maxiter, int use_precond, int *iters, double *eps, const double2 *pff_phi,
                  double2 *pff_out, int async_step)
{
    const Geometry &g = c->g;
    if (maxiter < 0) FAIL(c, 1, "maxiter < 0");
    CgGateHold gate_hold;  // smoqy_cg_gate: released on every return path
    // Speculation on the preconditioner's status record (round 3).  The record of the update_preconditioner! in front of this solve is
    // still on its way; what it can change for the launches below is (i) the count of multi-term frequencies (cheb_own_kernel's geometry)
    // and (ii) whether any walker is active at all.  Along a trajectory neither changes from solve to solve (the bounds move by more than
    // rbuf/2 a few times per run), so a solve that CAN be restarted — x === b: the right-hand side survives in its scratch copy and x
    // starts from zero — is launched on the host's current knowledge and checked at its first convergence poll, by which time the record
    // has landed in stream order.  If it says the launches were wrong in a way that matters (more chains than workgroups were given: those
    // frequencies were poisoned with NaN by the light workgroups; or a preconditioner became active while the plain path was running), the
    // solve starts over with the record consumed.  A stale count that is too LARGE, or walkers that became inactive, are harmless: the
    // kernels read orders and activation flags from device memory.  Warm-started solves and the very first solve of a handle wait as before.
    bool speculate = x_is_b && c->pstat_pending && c->pstat_ever;
restart:
    // The initial states go to the device from the page-locked template h_st0, which is rewritten only when (tol, maxiter, use_precond)
    // change — and then behind a stream synchronisation.  (h_st itself is the target of the polls; an asynchronous trajectory queues the
    // next solve's upload while this one's may not have run yet: rewriting the source in between zeroed tol / maxiter under a pending
    // copy — seen as spurious "unconverged" solves when the host ran far ahead of a small lattice.)
    if (!(c->st0_valid && c->st0_tol == tol && c->st0_maxiter == maxiter && c->st0_pre == (use_precond ? 1 : 0))) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int s = 0; s < g.nsys; ++s) {
            std::memset(&c->h_st0[s], 0, sizeof(CgState));
            c->h_st0[s].precond_on = use_precond ? 1 : 0;  // informational (the Chebyshev kernel reads each walker's `active` flag from the device)
            c->h_st0[s].tol = tol;
            c->h_st0[s].maxiter = maxiter;
        }
        c->st0_valid = true; c->st0_tol = tol; c->st0_maxiter = maxiter; c->st0_pre = use_precond ? 1 : 0;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_st, c->h_st0, (size_t)g.nsys * sizeof(CgState), hipMemcpyHostToDevice, c->stream));

    CgArgs a{};
    a.Lt = g.Lt; a.N = g.N; a.nsys = g.nsys; a.nrhs = g.nrhs; a.Tc = c->Tc; a.nchunk = c->nchunk;
    a.x = x; a.r = c->cg_r; a.p = c->cg_p; a.z = c->cg_z; a.th = c->d_th; a.b = b;
    a.v = c->cg_z;  // z = P⁻¹ r shares the buffer of A p: their lifetimes do not overlap
    a.part_pz = c->part_pz; a.part_rz = c->part_rz; a.part_rr = c->part_rr; a.part_bb = c->part_bb;
    a.st = c->d_st; a.tol = tol; a.maxiter = maxiter;
    a.rz_stride = 2 * g.Lt;
    if (pff_phi) { a.lam = c->d_lam; a.phi = pff_phi; a.x_out = pff_out; a.part_dot = c->part_c; }

    if (!x_is_b) {  // r0 = b - A x0  (ConjugateGradient.jl:119-120), in the twiddled basis
        launch_fft_twiddle(c->stream, x, c->d_th, g.Lt, g.N, g.nsys, 0);
        if (int rc = matvec_dev(c, SMOQY_OP_MTM, c->cg_z, x, nullptr, nullptr, 0, g.nsys, true)) return rc;
    }
    launch_cg_init(c->stream, a, x_is_b);  // needs neither of the two fields set below
    // Whether any walker's preconditioner is active — and the launch geometry of the Chebyshev kernel — is in the status record the last
    // update_preconditioner! sent after its kernels.  The host waits for it HERE, with the right-hand side's preparation and cg_init queued
    // behind those kernels, so the stream keeps working while the record travels (round 2 synchronised right after the Lanczos kernel).
    if (!speculate) if (int rc = pstat_wait(c)) return rc;
    bool any_pre = false;
    for (int w = 0; w < g.nw; ++w) any_pre = any_pre || (use_precond && c->pre[w].active);
    const int used_heavy = c->cheb_heavy;
    a.use_precond = any_pre ? 1 : 0;
    a.nrz = any_pre ? (cheb_split_active(kpm_args(c, nullptr, nullptr), c->kg) ? 2 * g.Lt : g.Lt) : c->nchunk;
    if (any_pre && c->tf_ok && c->use_tfft) {
        // fused iteration: the residual lives in frequency space from here on (r̂0 = FFT r0, in place), z0 = FFT⁻¹ P̂ r̂0 (:200)
        TfftArgs t = c->tf;
        t.src = c->cg_r; t.dst = c->cg_r; t.pre_tw = nullptr; t.post_tw = nullptr;
        launch_tfft(c->stream, 0, t);
        KpmArgs k = kpm_args(c, c->cg_r, nullptr);
        k.vout = c->cg_z;
        k.part_rz = c->part_rz;
        launch_cheb(c->stream, k, c->kg);
        c->cheb_name = cheb_kernel_name(k, c->kg);
        t.src = c->cg_z; t.dst = c->cg_z;
        launch_tfft(c->stream, 1, t);
    } else if (any_pre) if (int rc = precond_core(c, c->cg_r, c->cg_z, nullptr, c->part_rz)) return rc;  // z0 = P⁻¹ r0 (:200)
    launch_cg_start(c->stream, a);
    if (int rc = check_launch(c, "cg setup")) return rc;

    int launched = 0;
    bool finish_queued = false;  // cg_finish already sits behind the last burst (out-of-place form only)
    // smoqy_cg_split: fused path only (rocFFT plans and captured graphs cover the whole batch), and not while the fused-MᵀM launches are
    // being sampled for bench.py's roofline (the samples are of full-batch launches).  Automatic: two parts from 8 systems up (measured,
    // DESIGN.md §4.3).
    int nparts = c->cg_parts == 0 ? auto_parts(c) : c->cg_parts;
    if (nparts > g.nsys) nparts = g.nsys;
    if (!(any_pre && c->tf_ok && c->use_tfft) || c->use_graph || c->mvt.every != 0) nparts = 1;
    for (int q = 1; q < nparts; ++q)
        if (!c->part_stream[q - 1]) { nparts = q; break; }  // streams exist for the decided number of parts only
    int hint = 0, hslot = -1;
    for (int q = 0; q < 4; ++q)
        if (c->hint_tol[q] > 0 && std::fabs(std::log(c->hint_tol[q] / tol)) < 0.7) { hint = c->hint_iters[q]; hslot = q; }
    const bool async = async_step >= 0 && pff_phi != nullptr && x_is_b;
    while (launched < maxiter) {
        int burst = std::min(c->check_every, maxiter - launched);
        if (async) burst = std::min(maxiter, c->traj_hint[(size_t)async_step] + c->traj_margin);
        // first burst: one iteration MORE than the previous solve at this tolerance needed.  Consecutive solves of a trajectory differ by
        // at most an iteration or so; overshooting costs a few early-exit launches (≈ 1 µs each), a second poll costs ≈ 25 µs of idle stream
        if (!async && launched == 0 && hint + 1 > burst) burst = std::min(hint + 1, maxiter);
        hipGraphExec_t gexec = nullptr;
        if (c->use_graph) {
            // kGraphIters CG iterations captured once per (x, preconditioning, kernel configuration) and replayed:
            // the inner loop is launch bound at small batch (4 short dependent kernels per iteration)
            for (auto &gph : c->graphs)
                if (gph.exec && gph.epoch == c->graph_epoch && gph.x == (const void *)x && gph.pre == (int)any_pre && gph.Tc == c->Tc && gph.ffast == c->ff.enabled && gph.kfast == c->kg.fast) gexec = gph.exec;
            if (!gexec) {
                smoqy_ctx::IterGraph &slot = c->graphs[c->graph_next];
                c->graph_next = (c->graph_next + 1) % 4;
                if (slot.exec) { (void)hipGraphExecDestroy(slot.exec); slot.exec = nullptr; }
                if (slot.graph) { (void)hipGraphDestroy(slot.graph); slot.graph = nullptr; }
                hipError_t ge = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
                const char *stage = "hipStreamBeginCapture";
                bool ok = ge == hipSuccess;
                int rc = 0;
                if (ok) {
                    for (int q = 0; q < kGraphIters && rc == 0; ++q) rc = cg_iteration(c, a, any_pre);
                    ge = hipStreamEndCapture(c->stream, &slot.graph);
                    stage = rc ? "kernel launch during capture" : "hipStreamEndCapture";
                    ok = ge == hipSuccess && rc == 0 && slot.graph;
                }
                if (ok) { ge = hipGraphInstantiate(&slot.exec, slot.graph, nullptr, nullptr, 0); stage = "hipGraphInstantiate"; ok = ge == hipSuccess; }
                if (ok) {
                    slot.x = x; slot.pre = any_pre; slot.Tc = c->Tc; slot.ffast = c->ff.enabled; slot.kfast = c->kg.fast; slot.epoch = c->graph_epoch;
                    gexec = slot.exec;
                } else {
                    // not silent: the solve goes on with eager launches, graph replay is switched off, and the reason is kept where
                    // smoqy_cg_graph_status / smoqy_last_error can show it
                    char note[256];
                    snprintf(note, sizeof(note), "hipGraph capture of the CG iteration failed at %s (%s); falling back to eager launches", stage, hipGetErrorString(ge));
                    c->graph_note = note;
                    c->err = note;
                    (void)hipGetLastError();
                    if (slot.graph) { (void)hipGraphDestroy(slot.graph); slot.graph = nullptr; }
                    slot.exec = nullptr;
                    c->use_graph = 0;
                }
            }
        }
        if (gexec) {
            // whole graphs only: iterations past convergence or maxiter are workgroups that exit on their first load
            burst = ((burst + kGraphIters - 1) / kGraphIters) * kGraphIters;
            for (int it = 0; it < burst; it += kGraphIters) HIPCHK(c, hipGraphLaunch(gexec, c->stream));
        } else if (nparts > 1) {
            // multi-part pipeline: the systems are independent, so the iteration kernels of the first part run on the handle's stream and
            // those of the other parts on the handle's extra streams.  The chains drift out of phase, and one part's latency-bound Chebyshev
            // chain and load phases run under the other parts' bandwidth-bound kernels — the overlap that otherwise needs several handles on
            // several host threads.  Per system the same kernels run on a sub-range (a part of <= 8 systems selects the owner-computes MᵀM
            // kernel): results agree with the one-part form to rounding, bit for bit when the kernel family is the same.
            if (launched == 0) {
                HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));          // the set-up kernels above
                for (int q = 1; q < nparts; ++q) HIPCHK(c, hipStreamWaitEvent(c->part_stream[q - 1], c->ev_fork, 0));
            }
            for (int it = 0; it < burst; ++it)
                for (int q = 0; q < nparts; ++q) {
                    const int s0 = (int)((long)g.nsys * q / nparts), s1 = (int)((long)g.nsys * (q + 1) / nparts);
                    if (int rc = cg_iteration_fused(c, a, q == 0 ? c->stream : c->part_stream[q - 1], s0, s1 - s0)) {
                        // the part streams may still hold queued kernels that touch cg_r / cg_p / cg_z and the partial sums: drain them before
                        // the caller sees the error and reuses (or frees) those buffers on c->stream
                        for (int qq = 1; qq < nparts; ++qq) (void)hipStreamSynchronize(c->part_stream[qq - 1]);
                        return rc;
                    }
                }
            for (int q = 1; q < nparts; ++q) {
                HIPCHK(c, hipEventRecord(c->ev_part[q - 1], c->part_stream[q - 1]));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_part[q - 1], 0));  // the poll below (and everything after the solve) sees every part
            }
        } else {
            for (int it = 0; it < burst; ++it)
                if (int rc = cg_iteration(c, a, any_pre)) return rc;
        }
        launched += burst;
        // Ψ = Λ⁻¹Θᴴx̃ and the Φ·Ψ partials go out BEHIND EVERY BURST where they are written out of place (the PFFCalculator solve: x̃ stays
        // intact, the kernel may run any number of times): when the poll says "converged" the finish has already run, and the stream
        // does not idle between the host's wake-up and its next launch (round 4; 25 µs per solve in profiles/r03_gap_probe_1walker.txt)
        finish_queued = a.lam != nullptr;
        if (finish_queued) launch_cg_finish(c->stream, a);
        if (async) {  // the states of this solve stay on the device; nobody waits here
            HIPCHK(c, hipMemcpyAsync(c->d_traj_st + (size_t)async_step * g.nsys, c->d_st, (size_t)g.nsys * sizeof(CgState), hipMemcpyDeviceToDevice, c->stream));
            return check_launch(c, "cg loop (asynchronous)");
        }
        // (publishing the states into device-visible host memory and sleeping-then-spinning on a sequence number instead of this copy +
        // synchronisation was built and measured in round 4: one walker 25.9-28.0 -> 28.2-29.3 ms per sweep — not kept)
        HIPCHK(c, hipMemcpyAsync(c->h_st, c->d_st, (size_t)g.nsys * sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (int rc = check_launch(c, "cg loop")) return rc;
        if (speculate) {
            // the status record was queued in front of everything this solve launched: it has landed
            speculate = false;
            if (int rc = pstat_wait(c)) return rc;
            bool now_active = false;
            for (int w = 0; w < g.nw; ++w) now_active = now_active || (use_precond && c->pre[w].active);
            if (c->cheb_heavy > used_heavy || (now_active && !any_pre)) goto restart;  // the launches above were made on stale knowledge that mattered
        }
        bool all_done = true;
        for (int s = 0; s < g.nsys; ++s) all_done = all_done && (c->h_st[s].done != 0);
        if (all_done) break;
    }
    {   // remember how long this tolerance took
        int mx = 0;
        for (int s = 0; s < g.nsys; ++s) mx = std::max(mx, c->h_st[s].iters);
        if (hslot < 0) {
            hslot = 0;
            for (int q = 1; q < 4; ++q)
                if (c->hint_tol[q] == 0 || c->hint_iters[q] < c->hint_iters[hslot]) hslot = c->hint_tol[q] == 0 ? q : hslot;
            for (int q = 0; q < 4; ++q)
                if (c->hint_tol[q] == 0) { hslot = q; break; }
        }
        c->hint_tol[hslot] = tol;
        c->hint_iters[hslot] = mx;
    }
    if (!finish_queued) launch_cg_finish(c->stream, a);  // x = Θᴴ x̃ (asynchronous: whoever reads x next is ordered behind it on the stream)
    if (launched == 0) {
        // no poll has brought the state back yet (maxiter = 0): the convergence test of cg_start is all there is
        HIPCHK(c, hipMemcpyAsync(c->h_st, c->d_st, (size_t)g.nsys * sizeof(CgState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    // otherwise h_st already holds the final state: the last poll ran after the last launched iteration, and iterations past `done`
    // do not touch a system's state — a second read-back would only add a host synchronisation per solve
    if (int rc = check_launch(c, "cg finish")) return rc;
    for (int s = 0; s < g.nsys; ++s) {
        const CgState &st = c->h_st[s];
        if (!std::isfinite(st.eps)) FAIL(c, 7, "non-finite residual in CG for system %d (iters %d)", s, st.iters);
        if (iters) iters[s] = st.done == 1 ? st.iters : maxiter;  // (maxiter, ϵ) on non-convergence (:166 / :248)
        if (eps) eps[s] = st.eps;
    }
    return 0;
}

int smoqy_cg_solve_v(smoqy_ctx *c, int x, int b, double tol, int maxiter, int use_precond, int *iters, double *eps)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, x)) return rc;
    if (int rc = check_vec(c, b)) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (x == b) {
        // `x === b` (:112-116): r0 = b, x = 0.  b's buffer becomes x; a scratch copy serves as b.
        HIPCHK(c, hipMemcpyAsync(c->scr[0], c->vecs[b], c->vec_elems() * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
        return cg_dev(c, c->vecs[x], c->scr[0], true, tol, maxiter, use_precond, iters, eps);
    }
    return cg_dev(c, c->vecs[x], c->vecs[b], false, tol, maxiter, use_precond, iters, eps);
}

int smoqy_cg_solve(smoqy_ctx *c, void *x, const void *b, int x_is_b, int sys0, int count, double tol, int maxiter, int use_precond, int *iters, double *eps)
{
    CHECK_CTX(c);
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    const Geometry &g = c->g;
    if (count != g.nsys) {
        // systems outside the range get b = 0, which the start kernel retires immediately
        HIPCHK(c, hipMemsetAsync(c->scr[1], 0, c->vec_elems() * sizeof(double2), c->stream));
        HIPCHK(c, hipMemsetAsync(c->scr[2], 0, c->vec_elems() * sizeof(double2), c->stream));
    }
    if (int rc = upload_into(c, c->scr[1], b, sys0, count)) return rc;
    if (!x_is_b) if (int rc = upload_into(c, c->scr[2], x, sys0, count)) return rc;
    std::vector<int> it((size_t)g.nsys);
    std::vector<double> ep((size_t)g.nsys);
    if (int rc = cg_dev(c, c->scr[2], c->scr[1], x_is_b != 0, tol, maxiter, use_precond, it.data(), ep.data())) return rc;
    for (int k = 0; k < count; ++k) {
        if (iters) iters[k] = it[sys0 + k];
        if (eps) eps[k] = ep[sys0 + k];
    }
    return download_from(c, c->scr[2], x, sys0, count);
}


}  // extern "C"
