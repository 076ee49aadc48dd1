// C ABI of libsmoqy_hip.so (declared in include/smoqy_hip.h): handle management, boundary
// layout conversion, rocFFT plans, the KPM preconditioner's host-side bookkeeping and the
// on-device conjugate-gradient driver.  gfx950 / ROCm only; there is no CPU path.
#include <algorithm>
#include <time.h>
#include <chrono>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "smoqy_internal.h"

using namespace smoqy;

#include <condition_variable>

namespace {
std::string g_create_error;
std::once_flag g_rocfft_once;

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

struct WalkerPrecond {
    int active = 0;
    double emin = 0.0, emax = 0.0;
    std::vector<int> order;                  // nslot
    std::vector<std::vector<double2>> coefs; // nslot x order
    std::vector<double> lan_a, lan_b;
};

struct smoqy_ctx {
    Geometry g{};
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // second stream of the two-part CG pipeline (smoqy_cg_split): half the systems' iteration kernels run here, the other half's on `stream`
    static constexpr int kMaxParts = 4;
    hipStream_t part_stream[kMaxParts - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_part[kMaxParts - 1] = {nullptr, nullptr, nullptr};
    int cg_parts = 0;  // 0 = automatic, 1 = off, 2..4 = that many parts
    std::string err;
    int Tc = 1, nchunk = 1;
    bool user_Tc = false;
    // geometry
    int2 *d_bonds = nullptr;
    int *d_col_off = nullptr;
    // fields [nw][Lt][*]
    double *d_expV = nullptr, *d_ch = nullptr, *d_sh = nullptr, *d_lam = nullptr;
    double *d_shi = nullptr, *d_sbari = nullptr;  // T = ComplexF64 only: Im sinhΔτt [nw][Lt][Nh] and its tau-mean [nw][Nh]
    // user vectors
    std::vector<double2 *> vecs;
    // scratch
    double2 *d_stage = nullptr;     // nsys vectors in host layout
    // page-locked bounce arena for SMALL transfers whose host side is caller memory of unknown kind or a library temporary (pin_h2d /
    // pin_d2h): several handle threads never drive the runtime's own pageable-copy path at once
    char *h_pin = nullptr;
    size_t pin_cap = 0, pin_cur = 0;
    double *d_stage_real = nullptr; // max(N,Nh,Nph?) * Lt doubles (+ growth on demand)
    size_t stage_real_cap = 0;
    int *d_stage_int = nullptr;
    size_t stage_int_cap = 0;
    double2 *scr[3] = {nullptr, nullptr, nullptr};
    // cg
    double2 *cg_r = nullptr, *cg_p = nullptr, *cg_z = nullptr, *cg_v = nullptr;
    // lattices beyond the LDS limit (N > 2556): global staging area of the generic kernels, 4 N-vectors per workgroup
    double2 *d_big = nullptr;
    size_t big_stride = 0;
    double2 *part_pz = nullptr, *part_rz = nullptr, *part_c = nullptr, *d_dot_out = nullptr;
    double *part_rr = nullptr, *part_bb = nullptr;
    CgState *d_st = nullptr, *h_st = nullptr, *d_st_idle = nullptr;
    CgState *h_st0 = nullptr;  // page-locked template of the initial CG states (see cg_dev)
    bool st0_valid = false; double st0_tol = 0.0; int st0_maxiter = 0, st0_pre = 0;
    void *h_poll_dot = nullptr;  // pinned staging for per-system scalars (smoqy_pff_step_v)
    double2 *h_traj_dot = nullptr;  // pinned [Nt][nsys]: S_f of every step of a device trajectory, read once at its end
    double2 *d_traj_dot = nullptr;  // the same on the device: where the steps' dot_final kernels write
    size_t traj_cap = 0;
    int check_every = 4;
    // iterations the previous solve at (about) the same tolerance needed: consecutive solves of an HMC
    // trajectory converge in nearly the same number of iterations, so the first burst runs that far
    // before the host polls the device for the first time
    // captured CG iteration (hipGraph), keyed by everything baked into its kernel arguments
    // (x, preconditioning, kernel configuration) plus `epoch`: the kernel arguments captured in a graph hold d_coefs / maxorder /
    // nslot, the own-vs-rocFFT choice, Tc and the stream's rocFFT info BY VALUE, so every entry point that changes one of them
    // bumps graph_epoch (drop_graphs) and a stale graph is never replayed.
    struct IterGraph { const void *x = nullptr; int pre = 0, Tc = 0, ffast = 0, kfast = 0; unsigned epoch = 0; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
    IterGraph graphs[4];
    int graph_next = 0;
    unsigned graph_epoch = 1;
    std::string graph_note;  // why the last capture failed (also appended to smoqy_last_error)
    // Asynchronous trajectory (round 4, smoqy_hmc_async): the force solves of smoqy_hmc_trajectory_v are launched on the iteration counts the
    // PREVIOUS trajectory needed step by step (+ a margin), their states are kept per step on the device, and the host does not wait for any
    // of them until the trajectory's end, where every solve is checked (converged, finite); a miss restores x, p and runs the polling form.
    int traj_async = 1;
    std::vector<int> traj_hint;   // iterations step t of the last verified trajectory took (max over systems)
    double traj_hint_tol = 0.0;
    int traj_margin = 2;
    int traj_backoff = 0, traj_skip = 0;  // after a repeated trajectory the next traj_skip ones poll (doubling per consecutive miss, halving per success)
    CgState *d_traj_st = nullptr, *h_traj_st = nullptr;
    double *d_traj_save = nullptr;   // x and p at the start of the trajectory (the fall-back's starting point)
    char *d_traj_pre = nullptr;      // ... and the preconditioner's device state: accepted bounds, activation, orders, coefficients, status records
    size_t traj_pre_cap = 0;
    size_t traj_st_cap = 0, traj_save_cap = 0;
    long traj_async_runs = 0, traj_async_misses = 0;
    int wave_R = -1;             // run length of fdm_wave_kernel: -1 automatic (smoqy_matvec_wave)
    bool wave_off = false;
    const char *mtm_name = "";   // kernel family of the last full-batch fused MᵀM launch / Chebyshev launch (smoqy_describe)
    const char *cheb_name = "";
    // off by default: measured on MI355X the replay (≈10-16 µs per graph launch) does not beat six eager
    // launches per iteration (76.5 vs 80 ms per single-walker sweep); reset to 0 after a failed capture
    int use_graph = 0;
    double hint_tol[4] = {0, 0, 0, 0};
    int hint_iters[4] = {0, 0, 0, 0};
    // fft
    rocfft_plan plan_f = nullptr, plan_b = nullptr, plan_f_oop = nullptr;
    rocfft_execution_info fft_info = nullptr;
    void *fft_work = nullptr;
    double2 *d_tw = nullptr;  // theta_l / sqrt(Lt)  (unitary FourierTransformer)
    double2 *d_th = nullptr;  // theta_l
    double2 *d_wtab = nullptr;  // exp(-2 pi i q / Lt)
    int *d_tpos = nullptr;      // in-place tau-FFT: LDS row of each spectrum element
    TfftArgs tf{};            // plan of the own tau-FFT
    int tf_ok = 0, use_tfft = 1;
    int pstride = 0;          // per-system stride of the partial-sum arrays
    // kpm
    double rbuf = 0.10, a1 = 1.0, a2 = 1.0;
    int nlanczos = 20;
    int nslot = 0, maxorder = 64;
    std::vector<WalkerPrecond> pre;
    double *d_dbar = nullptr, *d_cbar = nullptr, *d_sbar = nullptr, *d_bounds = nullptr, *d_rand = nullptr, *d_lan = nullptr;
    double *d_rand_traj = nullptr;  // [Nt][nw][N (2N: complex T)] start vectors of a device trajectory (smoqy_hmc_trajectory_v)
    size_t rand_traj_cap = 0;
    int *d_order = nullptr, *d_active = nullptr;
    double2 *d_coefs = nullptr;
    KpmGeom kg{};
    FdmFast ff{};
    double2 *d_csf = nullptr;
    int *d_cs_varies = nullptr;
    int stream_R = -1;   // run length of the streaming MᵀM kernel: -1 automatic, 0 = chunked kernels only, >= 2 forced; smoqy_matvec_stream
    int cheb_heavy = 0;  // Sym cheb_own_kernel: number of leading frequency ranks with a multi-term expansion on any walker (upload_precond keeps it)
    std::vector<char> cs_const;  // [nw] 1 once the HOST has shown a walker's hoppings to be τ-independent (selects the one-pair-per-colour MᵀM kernel); 0 = unknown
    int2 *d_pbonds = nullptr, *d_psites = nullptr;
    int *d_pos = nullptr;
    int *d_poff = nullptr, *d_psrc = nullptr, *d_own = nullptr, *d_own_f = nullptr, *d_wave = nullptr, *d_fwave = nullptr;
    FdmWave fw{};
    double2 *d_pcs = nullptr;
    double *h_lan = nullptr;  // pinned [nw][2][1024]
    // device-resident bookkeeping of update_preconditioner! (PreUpd, kernels_kpm.hip): the host reads a 16-byte status record per
    // walker, and waits for it only where it needs what it says (the launch geometry of the Chebyshev kernel), with other work queued
    int *d_rebuild = nullptr, *d_pstat = nullptr, *h_pstat = nullptr;  // h_pstat pinned [nw][4]
    hipEvent_t ev_pstat = nullptr;
    bool pstat_pending = false;
    bool pstat_ever = false;     // a status record has been consumed at least once (the host's hints are meaningful)
    bool mirrors_stale = false;  // host copies of order / coefs / Lanczos coefficients are older than the device's (refreshed on demand by smoqy_precond_get*)
    // force terms
    struct ForceState {
        bool set = false;
        int Nph = 0, Nhol = 0, Nssh = 0, Q = 0;
        double dtau = 0;
        void *blob = nullptr;   // one device allocation holding every coupling table
        double *d_x = nullptr, *d_contrib = nullptr, *d_out = nullptr, *h_out = nullptr;
        double *d_bare = nullptr;  // [V⁰ (N) | t⁰ in checkerboard order (Nh)]
        bool bare_set = false, t_done = false;
        int t0_level = 1;  // 2: the bare hoppings are equal on every bond of a colour (set_cs_const level of walkers without SSH couplings)
        ForceArgs tmpl{};
        // EFA leapfrog: momenta, saved positions, per-(ω, mode) action eigenvalues and masses
        std::vector<int> finite_mass;
        bool efa_set = false;
        bool x0_valid = false;  // smoqy_efa_checkpoint(ctx, 0) has stored a checkpoint since smoqy_efa_config (restoring without one is refused)
        double *d_p = nullptr, *d_x0 = nullptr, *d_q = nullptr, *d_m = nullptr, *d_part = nullptr, *h_part = nullptr;
        int *d_fm = nullptr;
        int efa_SB = 8, efa_ntile = 0;
    } force;
    // GreensEstimator contractions
    struct GeState {
        bool set = false;
        int n_orb = 0, D = 0, Nc = 0;
        size_t n2 = 0;  // 2 Lτ · Nc, the size of one aperiodic array
        int Ld[2] = {1, 1};
        rocfft_plan fwd_sys = nullptr, inv_sys = nullptr, inv_w = nullptr;
        rocfft_execution_info info = nullptr;
        void *work = nullptr;
        double2 *A = nullptr, *B = nullptr, *P = nullptr, *out = nullptr;
        // four-point estimators: periodic (Lτ, L...) transforms over all pairs of one walker's random vectors
        size_t n1 = 0;  // Lτ · Nc
        int npairs = 0;
        rocfft_plan pfwd = nullptr, pinv = nullptr, pinv1 = nullptr;
        rocfft_execution_info pinfo = nullptr;
        void *pwork = nullptr;
        double2 *S[4] = {nullptr, nullptr, nullptr, nullptr}, *X = nullptr, *Y = nullptr, *tw[2] = {nullptr, nullptr};
        int2 *pairs = nullptr;
        double2 *bpart = nullptr, *bout = nullptr;  // boundary-term partial sums
    } ge;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // in-situ duration of the fused MᵀM launches of the CG loop (smoqy_matvec_timing): every `every`-th launch is
    // bracketed by an event pair on the handle's stream; read back after the timed region
    struct MvTiming {
        int every = 0, seen = 0, used = 0;
        std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
        unsigned long long *d_stamp = nullptr;  // [cap][wgs][2] start / end of every workgroup of each sampled launch (FdmArgs::stamp)
        int stamp_cap = 0, stamp_wgs = 0;
    } mvt;
    // event brackets around the four launches of fused CG iterations (smoqy_cg_iteration_timing): 5 events per sampled iteration
    struct IterTiming {
        int want = 0, used = 0;
        std::vector<hipEvent_t> ev;
    } itt;
    std::vector<int64_t> in_nt, in_cr;  // the neighbour table and colour ranges the handle was created from (smoqy_clone)

    size_t vec_elems() const { return (size_t)g.nsys * g.Lt * g.N; }
};

#define FAIL(ctx, code, ...)                                  \
    do {                                                      \
        char _b[512];                                         \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                \
        (ctx)->err = _b;                                      \
        return (code);                                        \
    } while (0)

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) FAIL(ctx, 2, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
    } while (0)

#define FFTCHK(ctx, expr)                                                                 \
    do {                                                                                  \
        rocfft_status _s = (expr);                                                        \
        if (_s != rocfft_status_success) FAIL(ctx, 3, "rocFFT error %d at %s:%d (%s)", (int)_s, __FILE__, __LINE__, #expr); \
    } while (0)

#define CHECK_CTX(ctx)            \
    if (!(ctx)) return 1;         \
    (void)hipSetDevice((ctx)->device)

// destroy every captured CG iteration: called whenever something baked into the captured kernel arguments changes
static void drop_graphs(smoqy_ctx *c)
{
    c->graph_epoch++;
    for (auto &gph : c->graphs) {
        if (gph.exec) { (void)hipGraphExecDestroy(gph.exec); gph.exec = nullptr; }
        if (gph.graph) { (void)hipGraphDestroy(gph.graph); gph.graph = nullptr; }
    }
}

// host-side proof that walker w's hoppings do (not) depend on τ; a change drops the captured CG graphs, which hold the kernel variant
// level: 0 unknown / τ-dependent, 1 τ-independent, 2 τ-independent AND the same (cosh, sinh) on every bond of a colour
static void set_cs_const(smoqy_ctx *c, int w, int level)
{
    if (c->cs_const.empty()) return;
    if (c->cs_const[(size_t)w] != (char)level) {
        c->cs_const[(size_t)w] = (char)level;
        drop_graphs(c);
    }
}
// 2 when v[h] is the same for all sorted bonds h of each colour, else 1 (v: one value per sorted bond, a τ-independent hopping table)
static int cs_level_of(const smoqy_ctx *c, const double *v, size_t stride)
{
    const Geometry &g = c->g;
    for (int col = 0; col < g.ncol; ++col) {
        const int h0 = (int)c->in_cr[2 * (size_t)col] - 1, h1 = (int)c->in_cr[2 * (size_t)col + 1];  // 1-based inclusive range
        for (int h = h0 + 1; h < h1; ++h)
            if (v[(size_t)h * stride] != v[(size_t)h0 * stride]) return 1;
    }
    return 2;
}

static int check_vec(smoqy_ctx *c, int id)
{
    if (id < 0 || id >= (int)c->vecs.size() || !c->vecs[id]) FAIL(c, 1, "invalid vector id %d", id);
    return 0;
}

static int check_launch(smoqy_ctx *c, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(c, 2, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
    return 0;
}

static void choose_chunking(smoqy_ctx *c)
{
    const Geometry &g = c->g;
    if (c->d_big) {  // global staging: one time slice per workgroup
        c->Tc = 1;
        c->nchunk = g.Lt;
        return;
    }
    if (!c->user_Tc) {
        // largest chunk that still gives >= 2 workgroups per CU and <= 64 KiB of LDS for the
        // fused MᵀM kernel; at small batch this degenerates to Tc = 1 (latency regime)
        int best = 1;
        const int cand[] = {2, 3, 4, 6, 8};
        for (int t : cand) {
            if (c->ff.enabled && t > 2) break;  // the register-resident kernels hold <= 3 slices
            const long wgs = (long)((g.Lt + t - 1) / t) * g.nsys;
            if (wgs >= 512 && fdm_lds_bytes(SMOQY_OP_MTM, g.N, t) <= 64 * 1024) best = t;
        }
        c->Tc = best;
    }
    c->nchunk = (g.Lt + c->Tc - 1) / c->Tc;
}

static FdmArgs fdm_args(smoqy_ctx *c, const double2 *in, double2 *out, double2 *partial, const CgState *cg, int sys0, int count)
{
    FdmArgs a{};
    const Geometry &g = c->g;
    a.Lt = g.Lt; a.N = g.N; a.Nh = g.Nh; a.ncol = g.ncol; a.nsys = g.nsys; a.nrhs = g.nrhs;
    a.Tc = c->Tc; a.nchunk = c->nchunk;
    a.bonds = c->d_bonds; a.col_off = c->d_col_off;
    a.expV = c->d_expV; a.ch = c->d_ch; a.sh = c->d_sh; a.shi = c->d_shi;
    a.in = in; a.out = out; a.partial = partial; a.cg = cg ? cg : c->d_st_idle;
    a.sys_first = sys0; a.sys_count = count;
    a.hop_re = 1.0; a.hop_im = 0.0; a.antiperiodic = 1;  // the reference operator
    a.scratch = c->d_big; a.scratch_stride = c->big_stride;
    return a;
}

static KpmArgs kpm_args(smoqy_ctx *c, double2 *v, const CgState *cg)
{
    KpmArgs k{};
    const Geometry &g = c->g;
    k.Lt = g.Lt; k.N = g.N; k.Nh = g.Nh; k.ncol = g.ncol; k.nsys = g.nsys; k.nrhs = g.nrhs; k.is_sym = g.is_sym;
    k.bonds = c->d_bonds; k.col_off = c->d_col_off;
    k.dbar = c->d_dbar; k.cbar = c->d_cbar; k.sbar = c->d_sbar; k.sbari = c->d_sbari;
    k.order = c->d_order; k.coefs = c->d_coefs; k.bounds = c->d_bounds; k.active = c->d_active;
    k.nslot = c->nslot; k.maxorder = c->maxorder;
    k.v = v; k.cg = cg ? cg : c->d_st_idle;
    k.part_rz = nullptr; k.rz_stride = 2 * g.Lt; k.scale = 1.0 / (double)g.Lt;  // two r·z slots per frequency: the component-split Chebyshev kernel fills both
    k.scratch = c->d_big; k.scratch_stride = c->big_stride;
    k.heavy = c->cheb_heavy; k.group = 8;  // light workgroups of cheb_own_kernel: eight single-term frequencies each
    return k;
}

static int ensure_stage_real(smoqy_ctx *c, size_t n)
{
    if (n <= c->stage_real_cap) return 0;
    if (c->d_stage_real) (void)hipFree(c->d_stage_real);
    c->d_stage_real = nullptr;
    HIPCHK(c, hipMalloc(&c->d_stage_real, n * sizeof(double)));
    c->stage_real_cap = n;
    return 0;
}

static int ensure_stage_int(smoqy_ctx *c, size_t n)
{
    if (n <= c->stage_int_cap) return 0;
    if (c->d_stage_int) (void)hipFree(c->d_stage_int);
    c->d_stage_int = nullptr;
    HIPCHK(c, hipMalloc(&c->d_stage_int, n * sizeof(int)));
    c->stage_int_cap = n;
    return 0;
}

// Small host -> device transfer through the handle's page-locked arena: the bytes are copied out of `src` before the call returns (the
// caller's buffer may be a temporary), the device copy is asynchronous on the handle's stream.  When the arena is full the stream is
// drained first — every earlier copy out of it has then landed — and the arena is reused from its start.
static int pin_reserve(smoqy_ctx *c, size_t bytes, char **slot)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (c->pin_cur + need > c->pin_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->pin_cur = 0;
        if (need > c->pin_cap) {
            if (c->h_pin) (void)hipHostFree(c->h_pin);
            c->h_pin = nullptr;
            c->pin_cap = 0;
            const size_t cap = std::max(2 * need, (size_t)1 << 20);  // room for the small transfers that follow a large one
            HIPCHK(c, hipHostMalloc((void **)&c->h_pin, cap, hipHostMallocDefault));
            c->pin_cap = cap;
        }
    }
    *slot = c->h_pin + c->pin_cur;
    c->pin_cur += need;
    return 0;
}

static int pin_h2d(smoqy_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    char *slot = nullptr;
    if (int rc = pin_reserve(c, bytes, &slot)) return rc;
    std::memcpy(slot, src, bytes);
    HIPCHK(c, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// small device -> host transfer into caller memory: lands in the arena, the stream is synchronised, then a plain memcpy
static int pin_d2h(smoqy_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    char *slot = nullptr;
    if (int rc = pin_reserve(c, bytes, &slot)) return rc;
    HIPCHK(c, hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(dst, slot, bytes);
    return 0;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

const char *smoqy_last_error(const smoqy_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

static void ge_release(smoqy_ctx *c)
{
    auto &G = c->ge;
    for (rocfft_plan p : {G.fwd_sys, G.inv_sys, G.inv_w, G.pfwd, G.pinv, G.pinv1})
        if (p) rocfft_plan_destroy(p);
    if (G.info) rocfft_execution_info_destroy(G.info);
    if (G.pinfo) rocfft_execution_info_destroy(G.pinfo);
    for (void *q : {G.work, (void *)G.A, (void *)G.B, (void *)G.P, (void *)G.out, G.pwork, (void *)G.S[0], (void *)G.S[1], (void *)G.S[2], (void *)G.S[3], (void *)G.X, (void *)G.Y, (void *)G.tw[0],
                    (void *)G.tw[1], (void *)G.pairs, (void *)G.bpart, (void *)G.bout})
        if (q) (void)hipFree(q);
    G = smoqy_ctx::GeState{};
}

int smoqy_destroy(smoqy_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto ps : c->part_stream)
        if (ps) (void)hipStreamSynchronize(ps);  // a solve that failed in mid-burst may have left kernels queued there: nothing is freed under them
    for (auto &gph : c->graphs) {
        if (gph.exec) (void)hipGraphExecDestroy(gph.exec);
        if (gph.graph) (void)hipGraphDestroy(gph.graph);
    }
    ge_release(c);
    if (c->plan_f) rocfft_plan_destroy(c->plan_f);
    if (c->plan_b) rocfft_plan_destroy(c->plan_b);
    if (c->plan_f_oop) rocfft_plan_destroy(c->plan_f_oop);
    if (c->fft_info) rocfft_execution_info_destroy(c->fft_info);
    void *ptrs[] = {c->d_bonds, c->d_col_off, c->d_expV, c->d_ch, c->d_sh, c->d_lam, c->d_stage, c->d_stage_real, c->d_stage_int, c->scr[0], c->scr[1], c->scr[2], c->cg_r, c->cg_p,
                    c->cg_z, c->cg_v, c->part_pz, c->part_rz, c->part_c, c->d_dot_out, c->part_rr, c->part_bb, c->d_st, c->d_st_idle, c->fft_work, c->d_tw, c->d_th, c->d_wtab, c->d_tpos, c->d_dbar, c->d_cbar, c->d_sbar, c->d_bounds,
                    c->d_rand, c->d_rand_traj, c->d_traj_dot, c->d_lan, c->d_order, c->d_active, c->d_coefs, c->d_pbonds, c->d_poff, c->d_psrc, c->d_pcs, c->d_csf, c->d_cs_varies, c->d_psites, c->d_pos, c->d_own, c->d_own_f == c->d_own ? nullptr : c->d_own_f, c->d_wave, c->d_fwave, c->d_big, c->d_shi, c->d_sbari};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (double2 *v : c->vecs)
        if (v) (void)hipFree(v);
    if (c->h_st) (void)hipHostFree(c->h_st);
    if (c->h_st0) (void)hipHostFree(c->h_st0);
    if (c->h_traj_st) (void)hipHostFree(c->h_traj_st);
    if (c->d_traj_st) (void)hipFree(c->d_traj_st);
    if (c->d_traj_save) (void)hipFree(c->d_traj_save);
    if (c->d_traj_pre) (void)hipFree(c->d_traj_pre);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_poll_dot) (void)hipHostFree(c->h_poll_dot);
    if (c->h_traj_dot) (void)hipHostFree(c->h_traj_dot);
    if (c->h_lan) (void)hipHostFree(c->h_lan);
    if (c->h_pstat) (void)hipHostFree(c->h_pstat);
    if (c->d_rebuild) (void)hipFree(c->d_rebuild);
    if (c->d_pstat) (void)hipFree(c->d_pstat);
    if (c->ev_pstat) (void)hipEventDestroy(c->ev_pstat);
    if (c->force.h_out) (void)hipHostFree(c->force.h_out);
    if (c->force.h_part) (void)hipHostFree(c->force.h_part);
    for (void *q : {c->force.blob, (void *)c->force.d_x, (void *)c->force.d_out, (void *)c->force.d_bare, (void *)c->force.d_p, (void *)c->force.d_x0, (void *)c->force.d_q,
                    (void *)c->force.d_m, (void *)c->force.d_part, (void *)c->force.d_fm})
        if (q) (void)hipFree(q);
    for (auto &e : c->mvt.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : c->itt.ev) (void)hipEventDestroy(e);
    if (c->mvt.d_stamp) (void)hipFree(c->mvt.d_stamp);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto e : c->ev_part) if (e) (void)hipEventDestroy(e);
    for (auto s : c->part_stream) if (s) (void)hipStreamDestroy(s);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return 0;
}

static int set_part_streams(smoqy_ctx *c, int nparts);
static int auto_parts(const smoqy_ctx *c);

// Stride of the per-slot coefficient table: the largest expansion order an ACTIVE preconditioner can ask for.  order = ⌊(ϵmax − ϵmin)(a1/ϕ + a2)⌋
// (KPMPreconditioner.jl:711) with 0 < ϵmin < 1 < ϵmax < 2 (:573) and ϕ ≥ π/Lτ (:220, folded :710), so order < 2 (a1 Lτ/π + a2).  Sizing the
// table for it once means the device can accept new bounds without the host growing anything (at Lτ = 128: 165 entries per slot).
static int coef_table_stride(const smoqy_ctx *c)
{
    const double a1 = c->g.is_sym ? 2.0 * c->a1 : c->a1;  // :263
    return (int)std::floor(2.0 * (a1 * c->g.Lt / M_PI + c->a2)) + 2;
}

// Lane program of cheb_wave_kernel (kernels_kpm_wave.hip): does the decomposition close into groups of four sites that a lane can own
// with the twice-applied colours inside its registers?  kind 1 — two colours, both perfect matchings, alternating along ONE cycle of
// N = 4·lanes sites (a ring): lane l owns r[4l … 4l+3].  kind 2 — four perfect matchings whose colours 1 and 2 close into 4-cycles
// s0 -c1- s1 -c2- s2 -c1- s3 -c2- s0 (plaquettes) labelled so that colour 0 pairs position p with position p^1 and colour 3 pairs p
// with 3-p of another plaquette, for EVERY site — the labelling is propagated from one plaquette and then verified in full; any
// violation means "no wave program" (kind 0) and the handle keeps cheb_own_kernel.  Table rows are documented at the kernel.
static void wave_program(int N, int ncol, const std::vector<int2> &pb, const std::vector<int> &psrc, const std::vector<int> &poff, const std::vector<std::vector<int>> &mate,
                         const std::vector<std::vector<int>> &bidx, std::vector<int> &tab, int &kind, int &lanes)
{
    kind = 0; lanes = 0;
    if ((ncol != 2 && ncol != 4) || N % 4 != 0 || N / 4 > 64 || N > 256) return;
    for (int col = 0; col < ncol; ++col)
        if (poff[col + 1] - poff[col] != N / 2) return;          // a padded list longer than N/2 holds self bonds
    for (size_t k = 0; k < psrc.size(); ++k)
        if (psrc[k] < 0 || pb[k].x == pb[k].y) return;
    const int n = N / 4;
    if (ncol == 2) {
        std::vector<int> ring((size_t)N), seen((size_t)N, 0);
        int s = pb[(size_t)poff[0]].x;
        for (int q = 0; q < N; ++q) {
            if (seen[s]) return;                                  // the cycle closed early: several rings
            seen[s] = 1; ring[q] = s;
            s = mate[q & 1][s];
        }
        if (s != ring[0]) return;
        tab.assign((size_t)11 * 64, 0);
        for (int l = 0; l < n; ++l) {
            const int *r = &ring[(size_t)4 * l];
            for (int p = 0; p < 4; ++p) tab[(size_t)p * 64 + l] = r[p];
            tab[4 * 64 + l] = bidx[0][r[0]]; tab[5 * 64 + l] = bidx[0][r[2]];
            tab[6 * 64 + l] = bidx[1][r[1]]; tab[7 * 64 + l] = bidx[1][r[3]]; tab[8 * 64 + l] = bidx[1][r[0]];
            tab[9 * 64 + l] = (l + 1) % n; tab[10 * 64 + l] = (l + n - 1) % n;
            if (mate[0][r[0]] != r[1] || mate[0][r[2]] != r[3] || mate[1][r[1]] != r[2] || mate[1][r[3]] != ring[(size_t)(4 * (l + 1)) % N] ||
                mate[1][r[0]] != ring[(size_t)(4 * l + N - 1) % N]) return;
        }
        kind = 1; lanes = n;
        return;
    }
    // plaquettes: label[site] = (plaquette, position), propagated breadth first through the colour-0 and colour-3 bonds
    std::vector<int> plq((size_t)N, -1), posn((size_t)N, -1), queue;
    std::vector<std::array<int, 4>> sites;
    auto place = [&](int t, int q) -> bool {  // a new plaquette with site t at position q; edge q -> q+1 is colour 1 for even q, colour 2 for odd q
        std::array<int, 4> s4{};
        int cur = t;
        for (int k = 0; k < 4; ++k) {
            const int p = (q + k) & 3;
            if (plq[cur] >= 0) return false;
            s4[(size_t)p] = cur;
            cur = mate[(p & 1) ? 2 : 1][cur];
        }
        if (cur != t) return false;                               // colours 1 and 2 do not close into a 4-cycle here
        const int id = (int)sites.size();
        for (int p = 0; p < 4; ++p) { plq[s4[(size_t)p]] = id; posn[s4[(size_t)p]] = p; }
        sites.push_back(s4);
        queue.push_back(id);
        return true;
    };
    if (!place(0, 0)) return;
    for (size_t h = 0; h < queue.size(); ++h) {
        const std::array<int, 4> s4 = sites[(size_t)queue[h]];
        for (int p = 0; p < 4; ++p) {
            const int t0 = mate[0][s4[(size_t)p]], t3 = mate[3][s4[(size_t)p]];
            if (plq[t0] < 0 && !place(t0, p ^ 1)) return;
            if (plq[t3] < 0 && !place(t3, 3 - p)) return;
        }
    }
    if ((int)sites.size() != n) return;                           // disconnected, or sites left over
    tab.assign((size_t)28 * 64, 0);
    for (int l = 0; l < n; ++l) {
        const std::array<int, 4> &s4 = sites[(size_t)l];
        for (int p = 0; p < 4; ++p) {
            const int s = s4[(size_t)p], t0 = mate[0][s], t3 = mate[3][s];
            if (plq[s] != l || posn[s] != p || posn[t0] != (p ^ 1) || posn[t3] != 3 - p || plq[t0] == l || plq[t3] == l) return;
            if (mate[(p & 1) ? 2 : 1][s] != s4[(size_t)((p + 1) & 3)]) return;
            tab[(size_t)p * 64 + l] = s;
            tab[(size_t)(8 + p) * 64 + l] = bidx[0][s];
            tab[(size_t)(12 + p) * 64 + l] = bidx[3][s];
            tab[(size_t)(16 + p) * 64 + l] = plq[t0];
            tab[(size_t)(20 + p) * 64 + l] = plq[t3];
            tab[(size_t)(24 + p) * 64 + l] = t0;
        }
        tab[4 * 64 + l] = bidx[1][s4[0]]; tab[5 * 64 + l] = bidx[1][s4[2]];
        tab[6 * 64 + l] = bidx[2][s4[1]]; tab[7 * 64 + l] = bidx[2][s4[3]];
    }
    kind = 2; lanes = n;
}

static int create_impl(smoqy_ctx *c, const int64_t *nt, const int64_t *cr)
{
    const Geometry &g = c->g;
    HIPCHK(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) FAIL(c, 4, "this library is built for gfx950 (MI355X) only; device %d is %s", c->device, prop.gcnArchName);
    {   // hipFuncSetAttribute applies to the CURRENT device only: one configuration pass (and one remembered error) per device ordinal,
        // made with that device current — a handle on a second GPU of the process gets the raised dynamic-LDS limit too
        constexpr int kMaxDev = 64;
        static std::mutex cfg_mu;
        static bool cfg_done[kMaxDev] = {};
        static hipError_t cfg_err[kMaxDev] = {};
        static const char *cfg_what[kMaxDev] = {};
        if (c->device < 0 || c->device >= kMaxDev) FAIL(c, 1, "device ordinal %d out of range", c->device);
        std::lock_guard<std::mutex> lk(cfg_mu);
        if (!cfg_done[c->device]) {
            cfg_err[c->device] = hipSuccess;
            cfg_what[c->device] = "";
            hipError_t (*cfgs[])(const char **) = {configure_fdm_kernels, configure_fdm_stream_kernels, configure_kpm_kernels, configure_tfft_kernels, configure_force_kernels};
            for (auto f : cfgs) {
                const char *w = "";
                const hipError_t e = f(&w);
                if (e != hipSuccess && cfg_err[c->device] == hipSuccess) { cfg_err[c->device] = e; cfg_what[c->device] = w; }
            }
            cfg_done[c->device] = true;
        }
        if (cfg_err[c->device] != hipSuccess)
            FAIL(c, 2, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for %s on device %d: %s", cfg_what[c->device], c->device, hipGetErrorString(cfg_err[c->device]));
    }
    HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    if (int rc = set_part_streams(c, std::min(auto_parts(c), g.nsys))) return rc;  // see set_part_streams for why here and not on first use
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));

    // neighbour table -> 0-based int2, colour offsets; validate the decomposition
    std::vector<int2> bonds((size_t)std::max(g.Nh, 1));
    for (int h = 0; h < g.Nh; ++h) {
        const int64_t i = nt[2 * h], j = nt[2 * h + 1];
        if (i < 1 || i > g.N || j < 1 || j > g.N || i == j) FAIL(c, 1, "neighbor_table column %d = (%lld, %lld) out of range 1..%d", h + 1, (long long)i, (long long)j, g.N);
        bonds[h] = make_int2((int)i - 1, (int)j - 1);
    }
    std::vector<int> off((size_t)g.ncol + 1, 0);
    int expect = 1;
    for (int col = 0; col < g.ncol; ++col) {
        const int64_t a = cr[2 * col], b = cr[2 * col + 1];
        if (a != expect || b < a || b > g.Nh) FAIL(c, 1, "color_ranges[%d] = %lld:%lld is not a contiguous partition of 1..%d", col + 1, (long long)a, (long long)b, g.Nh);
        off[col] = (int)a - 1;
        off[col + 1] = (int)b;
        expect = (int)b + 1;
        std::vector<char> seen((size_t)g.N, 0);
        for (int h = (int)a - 1; h < (int)b; ++h) {
            if (seen[bonds[h].x] || seen[bonds[h].y]) FAIL(c, 1, "colour %d is not a matching: bond %d shares a site with another bond of the same colour", col + 1, h + 1);
            seen[bonds[h].x] = seen[bonds[h].y] = 1;
        }
    }
    if (expect != g.Nh + 1) FAIL(c, 1, "color_ranges cover %d of %d bonds", expect - 1, g.Nh);
    HIPCHK(c, hipMalloc(&c->d_bonds, bonds.size() * sizeof(int2)));
    HIPCHK(c, hipMemcpy(c->d_bonds, bonds.data(), bonds.size() * sizeof(int2), hipMemcpyHostToDevice));
    HIPCHK(c, hipMalloc(&c->d_col_off, off.size() * sizeof(int)));
    HIPCHK(c, hipMemcpy(c->d_col_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));

    const size_t V = (size_t)g.Lt * g.N, VH = (size_t)g.Lt * std::max(g.Nh, 1);
    HIPCHK(c, hipMalloc(&c->d_expV, g.nw * V * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_ch, g.nw * VH * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_sh, g.nw * VH * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_lam, g.nw * V * sizeof(double)));
    HIPCHK(c, hipMemset(c->d_expV, 0, g.nw * V * sizeof(double)));
    HIPCHK(c, hipMemset(c->d_ch, 0, g.nw * VH * sizeof(double)));
    HIPCHK(c, hipMemset(c->d_sh, 0, g.nw * VH * sizeof(double)));
    HIPCHK(c, hipMemset(c->d_lam, 0, g.nw * V * sizeof(double)));
    if (g.is_cplx) {
        HIPCHK(c, hipMalloc(&c->d_shi, g.nw * VH * sizeof(double)));
        HIPCHK(c, hipMemset(c->d_shi, 0, g.nw * VH * sizeof(double)));
        HIPCHK(c, hipMalloc(&c->d_sbari, (size_t)g.nw * std::max(g.Nh, 1) * sizeof(double)));
        HIPCHK(c, hipMemset(c->d_sbari, 0, (size_t)g.nw * std::max(g.Nh, 1) * sizeof(double)));
    }

    const size_t ve = c->vec_elems();
    HIPCHK(c, hipMalloc(&c->d_stage, ve * sizeof(double2)));
    for (auto &s : c->scr) { HIPCHK(c, hipMalloc(&s, ve * sizeof(double2))); HIPCHK(c, hipMemset(s, 0, ve * sizeof(double2))); }
    double2 **cgv[] = {&c->cg_r, &c->cg_p, &c->cg_z, &c->cg_v};
    for (auto p : cgv) { HIPCHK(c, hipMalloc(p, ve * sizeof(double2))); HIPCHK(c, hipMemset(*p, 0, ve * sizeof(double2))); }
    c->pstride = std::max(2 * g.Lt, (g.N + 3) / 4);  // room for 2 Lt (per-frequency, per-component), nchunk and per-site-tile partials
    const size_t np = (size_t)g.nsys * c->pstride;
    HIPCHK(c, hipMalloc(&c->part_pz, np * sizeof(double2)));
    HIPCHK(c, hipMalloc(&c->part_rz, np * sizeof(double2)));
    HIPCHK(c, hipMalloc(&c->part_c, np * sizeof(double2)));
    HIPCHK(c, hipMalloc(&c->part_rr, np * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->part_bb, np * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_dot_out, (size_t)g.nsys * sizeof(double2)));
    HIPCHK(c, hipMalloc(&c->d_st, (size_t)g.nsys * sizeof(CgState)));
    HIPCHK(c, hipMemset(c->d_st, 0, (size_t)g.nsys * sizeof(CgState)));
    // an all-zero state ("nobody is done") for launches outside a CG loop: the kernels read the flag unconditionally — a load inside an
    // `if (cg)` is waited for on the spot, in front of everything else the workgroup could have asked for
    HIPCHK(c, hipMalloc(&c->d_st_idle, (size_t)g.nsys * sizeof(CgState)));
    HIPCHK(c, hipMemset(c->d_st_idle, 0, (size_t)g.nsys * sizeof(CgState)));
    HIPCHK(c, hipHostMalloc(&c->h_st, (size_t)g.nsys * sizeof(CgState)));
    HIPCHK(c, hipHostMalloc(&c->h_st0, (size_t)g.nsys * sizeof(CgState)));
    HIPCHK(c, hipHostMalloc(&c->h_poll_dot, (size_t)g.nsys * sizeof(double2)));
    choose_chunking(c);
    if (fdm_lds_bytes(SMOQY_OP_MTM, g.N, 1) > 160 * 1024 - 256) {
        // slices too large for LDS: the generic kernels stage them in global memory instead (one time slice per workgroup);
        // every workgroup of the largest launch (Lτ · nsys of them) gets room for four N-vectors
        c->big_stride = 4 * (size_t)g.N;
        HIPCHK(c, hipMalloc(&c->d_big, (size_t)g.Lt * g.nsys * c->big_stride * sizeof(double2)));
        c->Tc = 1;
        c->user_Tc = 1;
        c->nchunk = g.Lt;
    }

    // FourierTransformer: strided batched rocFFT along tau (stride nsys*N, distance 1)
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    HIPCHK(c, hipMalloc(&c->d_tw, (size_t)g.Lt * sizeof(double2)));
    HIPCHK(c, hipMalloc(&c->d_th, (size_t)g.Lt * sizeof(double2)));
    launch_make_twiddle(c->stream, c->d_tw, g.Lt, 1.0 / std::sqrt((double)g.Lt));
    launch_make_twiddle(c->stream, c->d_th, g.Lt, 1.0);
    {
        rocfft_plan_description desc = nullptr;
        FFTCHK(c, rocfft_plan_description_create(&desc));
        size_t stride[1] = {(size_t)g.nsys * g.N};
        FFTCHK(c, rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, stride, 1, 1, stride, 1));
        size_t len[1] = {(size_t)g.Lt};
        FFTCHK(c, rocfft_plan_create(&c->plan_f, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, (size_t)g.nsys * g.N, desc));
        FFTCHK(c, rocfft_plan_create(&c->plan_b, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_double, 1, len, (size_t)g.nsys * g.N, desc));
        FFTCHK(c, rocfft_plan_create(&c->plan_f_oop, rocfft_placement_notinplace, rocfft_transform_type_complex_forward, rocfft_precision_double, 1, len, (size_t)g.nsys * g.N, desc));
        rocfft_plan_description_destroy(desc);
        size_t wf = 0, wb = 0, wo = 0;
        FFTCHK(c, rocfft_plan_get_work_buffer_size(c->plan_f_oop, &wo));
        FFTCHK(c, rocfft_plan_get_work_buffer_size(c->plan_f, &wf));
        FFTCHK(c, rocfft_plan_get_work_buffer_size(c->plan_b, &wb));
        FFTCHK(c, rocfft_execution_info_create(&c->fft_info));
        const size_t wsz = std::max(std::max(wf, wb), wo);
        if (wsz) {
            HIPCHK(c, hipMalloc(&c->fft_work, wsz));
            FFTCHK(c, rocfft_execution_info_set_work_buffer(c->fft_info, c->fft_work, wsz));
        }
        FFTCHK(c, rocfft_execution_info_set_stream(c->fft_info, c->stream));
    }

    {   // own tau-FFT (kernels_tfft.hip) when Lt factors into 2, 3, 5, 7
        c->tf_ok = tfft_plan(g.Lt, g.N, c->tf) ? 1 : 0;
        c->tf.nsys = g.nsys;
        std::vector<double2> wt((size_t)g.Lt);
        for (int q = 0; q < g.Lt; ++q) wt[q] = make_double2(std::cos(2.0 * M_PI * q / g.Lt), -std::sin(2.0 * M_PI * q / g.Lt));
        HIPCHK(c, hipMalloc(&c->d_wtab, wt.size() * sizeof(double2)));
        HIPCHK(c, hipMemcpy(c->d_wtab, wt.data(), wt.size() * sizeof(double2), hipMemcpyHostToDevice));
        c->tf.wtab = c->d_wtab;
        if (c->tf_ok) {
            std::vector<int> pos((size_t)g.Lt);
            tfft_positions(c->tf, pos.data());
            HIPCHK(c, hipMalloc(&c->d_tpos, pos.size() * sizeof(int)));
            HIPCHK(c, hipMemcpy(c->d_tpos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
            c->tf.pos = c->d_tpos;
            // form of the own tau-FFT when nobody chose one (smoqy_tfft_form, SMOQY_TFFT_SLIM): in place from 32 systems per launch — at that
            // size the launches are HBM bound and six workgroups per CU beat four (64 walkers on one stream: 253.7 -> 242.9 ms per sweep,
            // 128: 491 -> 464), below it the two-image form's fewer passes win (16 walkers: DESIGN.md §4.3)
            static const bool form_free = tuning_env(kTuneTfftSlim) < 0;
            if (form_free && c->tf.slim_ok && g.nsys >= 32) c->tf.slim = 1;
        }
    }

    // KPM preconditioner state
    c->nslot = g.is_sym ? (g.Lt + 1) / 2 : g.Lt;  // KPMPreconditioner.jl:254-257, 268-271
    c->pre.resize((size_t)g.nw);
    for (auto &p : c->pre) { p.order.assign((size_t)c->nslot, 0); p.coefs.resize((size_t)c->nslot); }
    HIPCHK(c, hipMalloc(&c->d_dbar, (size_t)g.nw * g.N * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_cbar, (size_t)g.nw * std::max(g.Nh, 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_sbar, (size_t)g.nw * std::max(g.Nh, 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_bounds, (size_t)g.nw * 2 * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_rand, (size_t)g.nw * g.N * (g.is_cplx ? 2 : 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_lan, (size_t)g.nw * 2 * 1024 * sizeof(double)));
    HIPCHK(c, hipHostMalloc(&c->h_lan, (size_t)g.nw * 2 * 1024 * sizeof(double)));
    {
        // padded per-colour bond lists for the register-resident KPM kernels (kernels_kpm.hip)
        std::vector<int2> pb;
        std::vector<int> psrc, poff((size_t)g.ncol + 1, 0);
        int maxp = 0;
        for (int col = 0; col < g.ncol; ++col) {
            std::vector<char> seen((size_t)g.N, 0);
            poff[col] = (int)pb.size();
            for (int h = off[col]; h < off[col + 1]; ++h) {
                pb.push_back(bonds[h]);
                psrc.push_back(h);
                seen[bonds[h].x] = seen[bonds[h].y] = 1;
            }
            for (int i = 0; i < g.N; ++i)
                if (!seen[i]) { pb.push_back(make_int2(i, i)); psrc.push_back(-1); }
            poff[col + 1] = (int)pb.size();
            maxp = std::max(maxp, poff[col + 1] - poff[col]);
        }
        c->kg.ptotal = (int)pb.size();
        c->kg.threads = std::max(64, ((maxp + 63) / 64) * 64);
        c->kg.fast = (!g.is_cplx && g.ncol >= 1 && g.ncol <= kMaxColours && maxp <= 1024) ? 1 : 0;  // complex hoppings: generic kernels
        // LDS positions: the first colour's x sites in list order, then its y sites
        std::vector<int> pos((size_t)g.N);
        for (int i = 0; i < g.N; ++i) pos[i] = i;
        if (g.ncol >= 1) {
            int q = 0;
            for (int k = poff[0]; k < poff[1]; ++k) pos[pb[k].x] = q++;
            for (int k = poff[0]; k < poff[1]; ++k)
                if (pb[k].y != pb[k].x) pos[pb[k].y] = q++;
            if (q != g.N) FAIL(c, 8, "internal: first colour's padded list does not cover all sites (%d of %d)", q, g.N);
        }
        std::vector<int2> pbpos(pb.size());
        for (size_t k = 0; k < pb.size(); ++k) pbpos[k] = make_int2(pos[pb[k].x], pos[pb[k].y]);
        HIPCHK(c, hipMalloc(&c->d_psites, std::max<size_t>(pb.size(), 1) * sizeof(int2)));
        HIPCHK(c, hipMalloc(&c->d_pos, (size_t)g.N * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->d_pos, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!pb.empty()) HIPCHK(c, hipMemcpy(c->d_psites, pb.data(), pb.size() * sizeof(int2), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc(&c->d_pbonds, std::max<size_t>(pb.size(), 1) * sizeof(int2)));
        HIPCHK(c, hipMalloc(&c->d_psrc, std::max<size_t>(pb.size(), 1) * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->d_poff, poff.size() * sizeof(int)));
        HIPCHK(c, hipMalloc(&c->d_pcs, (size_t)g.nw * std::max<size_t>(pb.size(), 1) * sizeof(double2)));
        if (!pb.empty()) {
            HIPCHK(c, hipMemcpy(c->d_pbonds, pbpos.data(), pbpos.size() * sizeof(int2), hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->d_psrc, psrc.data(), psrc.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        HIPCHK(c, hipMemcpy(c->d_poff, poff.data(), poff.size() * sizeof(int), hipMemcpyHostToDevice));
        c->kg.psites = c->d_psites; c->kg.pos = c->d_pos;
        c->kg.pbonds = c->d_pbonds; c->kg.poff = c->d_poff; c->kg.psrc = c->d_psrc; c->kg.pcs = c->d_pcs;
        if (c->kg.fast) {
            // owner-computes tables: the owned colour is one that is applied twice (per Chebyshev step in
            // cheb_own_kernel, per B apply in fdm_own_kernel), so that its two stages need no exchange at all
            const int T = c->kg.threads;
            std::vector<std::vector<int>> mate((size_t)g.ncol, std::vector<int>((size_t)g.N)), bidx((size_t)g.ncol, std::vector<int>((size_t)g.N));
            for (int col = 0; col < g.ncol; ++col)
                for (int k = poff[col]; k < poff[col + 1]; ++k) {
                    mate[col][pb[k].x] = pb[k].y; mate[col][pb[k].y] = pb[k].x;
                    bidx[col][pb[k].x] = bidx[col][pb[k].y] = k;
                }
            auto build_own = [&](int q, int **dptr, int *nb_out) -> int {
                const int nb = poff[q + 1] - poff[q];
                std::vector<int> slot((size_t)g.N, 0);
                for (int j = 0; j < nb; ++j) {
                    const int2 b = pb[(size_t)poff[q] + j];
                    slot[b.x] = j;
                    if (b.y != b.x) slot[b.y] = T + j;
                }
                std::vector<int> own((size_t)(4 + 4 * g.ncol) * T, 0);
                for (int j = 0; j < nb; ++j) {
                    const int2 b = pb[(size_t)poff[q] + j];
                    own[0 * (size_t)T + j] = b.x; own[1 * (size_t)T + j] = b.y;
                    own[2 * (size_t)T + j] = mate[0][b.x]; own[3 * (size_t)T + j] = mate[0][b.y];
                    for (int col = 0; col < g.ncol; ++col) {
                        own[(size_t)(4 + 4 * col + 0) * T + j] = slot[mate[col][b.x]];
                        own[(size_t)(4 + 4 * col + 1) * T + j] = slot[mate[col][b.y]];
                        own[(size_t)(4 + 4 * col + 2) * T + j] = bidx[col][b.x];
                        own[(size_t)(4 + 4 * col + 3) * T + j] = bidx[col][b.y];
                    }
                }
                HIPCHK(c, hipMalloc(dptr, own.size() * sizeof(int)));
                HIPCHK(c, hipMemcpy(*dptr, own.data(), own.size() * sizeof(int), hipMemcpyHostToDevice));
                *nb_out = nb;
                return 0;
            };
            const int q_cheb = g.ncol >= 3 ? 1 : 0, q_fdm = g.ncol >= 2 ? 1 : 0;
            if (int rc = build_own(q_cheb, &c->d_own, &c->kg.own_n)) return rc;
            c->kg.own = c->d_own; c->kg.own_q = q_cheb;
            {   // is the colour-0 exchange of the Chebyshev lane program wave-local (KpmGeom::wl0)?  Lane j holds site b.x in slot j and b.y
                // in slot T + j; the mate of b.x must sit in a second slot and the mate of b.y in a first slot of the same 64-lane wavefront
                const int nb = poff[q_cheb + 1] - poff[q_cheb];
                std::vector<int> slot((size_t)g.N, 0);
                for (int j = 0; j < nb; ++j) {
                    const int2 b = pb[(size_t)poff[q_cheb] + j];
                    slot[b.x] = j;
                    if (b.y != b.x) slot[b.y] = T + j;
                }
                bool ok = g.ncol >= 3;
                // ... and, stronger: are they the lane's neighbours in its row of 16 lanes, cyclically (mate of the first site one lane
                // down, mate of the second one lane up, or the other way round)?  Then the exchange is two DPP row rotations — register
                // moves, no LDS permute unit at all (wl0 = 2 / 3).  Honeycomb L = 16: the colour-0 bond is the one along the rows of 16 cells.
                bool rot_dn = ok && nb == T && T % 16 == 0, rot_up = rot_dn;
                for (int j = 0; j < nb && ok; ++j) {
                    const int2 b = pb[(size_t)poff[q_cheb] + j];
                    const int sx = slot[mate[0][b.x]], sy = slot[mate[0][b.y]];
                    ok = sx >= T && (sx - T) / 64 == j / 64 && sy < T && sy / 64 == j / 64;
                    const int dn = (j & ~15) | ((j - 1) & 15), up = (j & ~15) | ((j + 1) & 15);
                    rot_dn = rot_dn && ok && sx - T == dn && sy == up;
                    rot_up = rot_up && ok && sx - T == up && sy == dn;
                }
                c->kg.wl0 = ok ? (rot_dn ? 2 : (rot_up ? 3 : 1)) : 0;
            }
            // the MᵀM kernel for small launches shares the lane layout when it owns the same colour: the DPP form of the colour-0 exchange too
            static const bool wl_env_off = tuning_env(kTuneChebWl0) == 0 || tuning_env(kTuneChebWl0) == 1;
            c->ff.wl0 = (q_fdm == q_cheb && c->kg.wl0 >= 2 && !wl_env_off) ? c->kg.wl0 : 0;
            {   // one-wavefront-per-chain lane program of the Sym Chebyshev kernel (kernels_kpm_wave.hip), where the lattice has one
                std::vector<int> tab;
                int kind = 0, lanes = 0;
                wave_program(g.N, g.ncol, pb, psrc, poff, mate, bidx, tab, kind, lanes);
                if (kind) {
                    HIPCHK(c, hipMalloc(&c->d_wave, tab.size() * sizeof(int)));
                    HIPCHK(c, hipMemcpy(c->d_wave, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
                    c->kg.wave = c->d_wave; c->kg.wave_kind = kind; c->kg.wave_lanes = lanes;
                }
            }
            {   // lane program of the one-wavefront-per-run MᵀM kernel (kernels_fdm_wave.hip): every colour a perfect matching, real hoppings
                bool perfect = !g.is_cplx && g.ncol <= kFdmColours;
                for (int col = 0; col < g.ncol && perfect; ++col) perfect = poff[col + 1] - poff[col] == g.N / 2 && g.N % 2 == 0;
                for (size_t k = 0; k < psrc.size() && perfect; ++k) perfect = psrc[k] >= 0 && pb[k].x != pb[k].y;
                if (perfect) {
                    std::vector<int> tab;
                    int kind = 0, lanes = 0;
                    bool rot = false;
                    fdm_wave_program(g.N, g.ncol, mate, bidx, tab, kind, lanes, rot);
                    if (kind) {
                        HIPCHK(c, hipMalloc(&c->d_fwave, tab.size() * sizeof(int)));
                        HIPCHK(c, hipMemcpy(c->d_fwave, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
                        c->fw.tab = c->d_fwave; c->fw.kind = kind; c->fw.lanes = lanes; c->fw.rot = rot ? 1 : 0;
                    }
                }
            }
            if (q_fdm == q_cheb) { c->d_own_f = c->d_own; c->ff.own_n = c->kg.own_n; }
            else if (int rc = build_own(q_fdm, &c->d_own_f, &c->ff.own_n)) return rc;
            c->ff.own = c->d_own_f;
        }
        HIPCHK(c, hipMalloc(&c->d_csf, (size_t)g.nw * g.Lt * std::max<size_t>(pb.size(), 1) * sizeof(double2)));
        HIPCHK(c, hipMemset(c->d_csf, 0, (size_t)g.nw * g.Lt * std::max<size_t>(pb.size(), 1) * sizeof(double2)));
        HIPCHK(c, hipMalloc(&c->d_cs_varies, (size_t)g.nw * sizeof(int)));
        HIPCHK(c, hipMemset(c->d_cs_varies, 0, (size_t)g.nw * sizeof(int)));
        c->ff.cs_varies = c->d_cs_varies;
        c->cs_const.assign((size_t)g.nw, 0);
        c->ff.psites = c->d_psites; c->ff.pos = c->d_pos;
        c->ff.pbonds = c->d_pbonds; c->ff.poff = c->d_poff; c->ff.csf = c->d_csf; c->ff.ptotal = c->kg.ptotal; c->ff.threads = c->kg.threads;
        c->ff.enabled = (!g.is_cplx && g.ncol >= 1 && g.ncol <= kFdmColours && maxp <= 1024) ? 1 : 0;  // Sym: fdm_fast/own kernels; Asym: fdm_fast_asym_kernel
        {   // FdmFast::full: no padded self bond anywhere and every list exactly one bond per lane
            bool full = g.ncol >= 1 && g.N == 2 * c->kg.threads;
            for (int col = 0; col < g.ncol && full; ++col) full = poff[col + 1] - poff[col] == c->kg.threads;
            for (size_t k = 0; k < psrc.size() && full; ++k) full = psrc[k] >= 0;
            c->ff.full = full ? 1 : 0;
        }
        choose_chunking(c);
    }
    HIPCHK(c, hipMalloc(&c->d_order, (size_t)g.nw * c->nslot * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->d_active, (size_t)g.nw * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_active, 0, (size_t)g.nw * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_order, 0, (size_t)g.nw * c->nslot * sizeof(int)));
    c->maxorder = std::max(c->maxorder, coef_table_stride(c));
    HIPCHK(c, hipMalloc(&c->d_coefs, (size_t)g.nw * c->nslot * c->maxorder * sizeof(double2)));
    HIPCHK(c, hipMemset(c->d_coefs, 0, (size_t)g.nw * c->nslot * c->maxorder * sizeof(double2)));
    HIPCHK(c, hipMemset(c->d_bounds, 0, (size_t)g.nw * 2 * sizeof(double)));
    HIPCHK(c, hipMalloc(&c->d_rebuild, (size_t)g.nw * sizeof(int)));
    HIPCHK(c, hipMalloc(&c->d_pstat, (size_t)g.nw * 4 * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_rebuild, 0, (size_t)g.nw * sizeof(int)));
    HIPCHK(c, hipMemset(c->d_pstat, 0, (size_t)g.nw * 4 * sizeof(int)));
    HIPCHK(c, hipHostMalloc((void **)&c->h_pstat, (size_t)g.nw * 4 * sizeof(int), hipHostMallocDefault));
    std::memset(c->h_pstat, 0, (size_t)g.nw * 4 * sizeof(int));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pstat, hipEventDisableTiming));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "create");
}

int smoqy_create(smoqy_ctx **out, int Ltau, int N, int Nh, int ncolors, const int64_t *neighbor_table, const int64_t *color_ranges, int is_sym, int is_complex_T, int nwalkers, int nrhs, int device_id)
{
    if (!out) return 1;
    *out = nullptr;
    if (Ltau < 1 || N < 1 || Nh < 0 || ncolors < 0 || nwalkers < 1 || nrhs < 1 || (Nh > 0 && (!neighbor_table || !color_ranges))) {
        g_create_error = "smoqy_create: invalid dimensions or null tables";
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        g_create_error = "smoqy_create: no HIP device visible — this library has no CPU path";
        return 4;
    }
    smoqy_ctx *c = new smoqy_ctx();
    c->g = Geometry{Ltau, N, Nh, ncolors, nwalkers, nrhs, nwalkers * nrhs, is_sym ? 1 : 0, is_complex_T ? 1 : 0};
    if (device_id < 0) { if (hipGetDevice(&c->device) != hipSuccess) c->device = 0; }
    else c->device = device_id;
    int rc = create_impl(c, neighbor_table, color_ranges);
    if (rc) {
        g_create_error = c->err;
        smoqy_destroy(c);
        return rc;
    }
    if (Nh > 0) {
        c->in_nt.assign(neighbor_table, neighbor_table + 2 * (size_t)Nh);
        c->in_cr.assign(color_ranges, color_ranges + 2 * (size_t)ncolors);
    }
    *out = c;
    return 0;
}

// a second handle on the same lattice, propagator form, device and preconditioner configuration with `nrhs` right-hand sides per walker
// (e.g. the GreensEstimator's follower handle: Nrv systems per walker, fields copied over with smoqy_copy_fields)
int smoqy_clone(smoqy_ctx **out, const smoqy_ctx *src, int nrhs)
{
    if (!out || !src || nrhs < 1) { g_create_error = "smoqy_clone: null handle or nrhs < 1"; return 1; }
    const Geometry &g = src->g;
    if (int rc = smoqy_create(out, g.Lt, g.N, g.Nh, g.ncol, src->in_nt.data(), src->in_cr.data(), g.is_sym, g.is_cplx, g.nw, nrhs, src->device)) return rc;
    const int rc = smoqy_precond_config(*out, src->rbuf, src->nlanczos, src->a1, src->a2);
    if (rc) {  // do not hand back (or leak) a half-configured handle
        g_create_error = std::string("smoqy_clone: ") + (*out)->err;
        smoqy_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int smoqy_set_stream(smoqy_ctx *c, void *s)
{
    CHECK_CTX(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto ps : c->part_stream) if (ps) HIPCHK(c, hipStreamSynchronize(ps));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    FFTCHK(c, rocfft_execution_info_set_stream(c->fft_info, c->stream));
    drop_graphs(c);
    return 0;
}

// page-locked host memory for arrays that cross the boundary repeatedly (fields, phonon positions):
// transfers from it run at full PCIe rate
int smoqy_host_alloc(smoqy_ctx *c, void **ptr, size_t bytes)
{
    CHECK_CTX(c);
    HIPCHK(c, hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return 0;
}

int smoqy_host_free(smoqy_ctx *c, void *ptr)
{
    CHECK_CTX(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipHostFree(ptr));
    return 0;
}

// page-lock memory the caller already owns (a Julia array, a shared-memory segment): same effect as smoqy_host_alloc for transfers from it
int smoqy_host_register(smoqy_ctx *c, void *ptr, size_t bytes)
{
    CHECK_CTX(c);
    if (!ptr || bytes == 0) FAIL(c, 1, "smoqy_host_register: null pointer or zero size");
    HIPCHK(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return 0;
}

int smoqy_host_unregister(smoqy_ctx *c, void *ptr)
{
    CHECK_CTX(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipHostUnregister(ptr));
    return 0;
}

int smoqy_sync(smoqy_ctx *c)
{
    CHECK_CTX(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int smoqy_dims(const smoqy_ctx *c, int d[6])
{
    CHECK_CTX(c);
    d[0] = c->g.Lt; d[1] = c->g.N; d[2] = c->g.Nh; d[3] = c->g.ncol; d[4] = c->g.nw; d[5] = c->g.nrhs;
    return 0;
}

int smoqy_traits(const smoqy_ctx *c, int t[8])
{
    CHECK_CTX(c);
    if (!t) return 1;
    t[0] = c->g.is_sym; t[1] = c->g.is_cplx; t[2] = c->kg.fast; t[3] = c->kg.wl0; t[4] = c->kg.wave_kind; t[5] = c->kg.wave_lanes; t[6] = c->ff.enabled; t[7] = c->ff.full;
    return 0;
}

int smoqy_describe(const smoqy_ctx *c, char *buf, size_t n)
{
    CHECK_CTX(c);
    if (!buf || n == 0) return 1;
    snprintf(buf, n, "{\"mtm\": \"%s\", \"cheb\": \"%s\", \"tfft\": \"%s\"}", c->mtm_name, c->cheb_name,
             !c->tf_ok ? "rocFFT + cg_update kernels" : (c->tf.slim ? "tfft_kernel (in-place form)" : "tfft_kernel (two-image form)"));
    return 0;
}

int smoqy_set_tau_chunk(smoqy_ctx *c, int Tc)
{
    CHECK_CTX(c);
    drop_graphs(c);
    if (Tc <= 0) { c->user_Tc = false; choose_chunking(c); return 0; }
    if (c->d_big) {
        if (Tc != 1) FAIL(c, 1, "lattices beyond the LDS limit run with one time slice per workgroup");
        return 0;
    }
    if (fdm_lds_bytes(SMOQY_OP_MTM, c->g.N, Tc) > 160 * 1024 - 256) FAIL(c, 1, "tau chunk %d needs more than 160 KiB of LDS at N = %d", Tc, c->g.N);
    c->user_Tc = true;
    c->Tc = std::min(Tc, c->g.Lt);
    choose_chunking(c);
    return 0;
}

int smoqy_get_tau_chunk(const smoqy_ctx *c, int *Tc)
{
    CHECK_CTX(c);
    *Tc = c->Tc;
    return 0;
}

// ---- fields -----------------------------------------------------------------------------------

static int upload_real_field(smoqy_ctx *c, const double *host, double *dev, int n)
{
    const size_t cnt = (size_t)c->g.Lt * n;
    if (cnt == 0) return 0;
    if (int rc = ensure_stage_real(c, cnt)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_stage_real, host, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_transpose_real_in(c->stream, c->d_stage_real, dev, c->g.Lt, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next field
    return 0;
}

static int download_real_field(smoqy_ctx *c, const double *dev, double *host, int n)
{
    const size_t cnt = (size_t)c->g.Lt * n;
    if (cnt == 0) return 0;
    if (int rc = ensure_stage_real(c, cnt)) return rc;
    launch_transpose_real_out(c->stream, dev, c->d_stage_real, c->g.Lt, n);
    HIPCHK(c, hipMemcpyAsync(host, c->d_stage_real, cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

#define CHECK_WALKER(c, w) \
    if ((w) < 0 || (w) >= (c)->g.nw) FAIL(c, 1, "walker %d out of range 0..%d", (w), (c)->g.nw - 1)

int smoqy_update_fields(smoqy_ctx *c, int w, const double *expV, const double *ch, const double *sh)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    const Geometry &g = c->g;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = upload_real_field(c, expV, c->d_expV + (size_t)w * g.Lt * g.N, g.N)) return rc;
    if (g.is_cplx) {
        // T = ComplexF64: coshΔτt and sinhΔτt arrive as complex128 arrays (the reference stores both as Matrix{T}); the device keeps
        // Re cosh, Re sinh and Im sinh as separate real arrays
        const size_t cnt = (size_t)g.Lt * g.Nh;
        std::vector<double> re(cnt), im(cnt);
        for (size_t k = 0; k < cnt; ++k) re[k] = ch[2 * k];
        if (int rc = upload_real_field(c, re.data(), c->d_ch + (size_t)w * cnt, g.Nh)) return rc;
        for (size_t k = 0; k < cnt; ++k) { re[k] = sh[2 * k]; im[k] = sh[2 * k + 1]; }
        if (int rc = upload_real_field(c, re.data(), c->d_sh + (size_t)w * cnt, g.Nh)) return rc;
        if (int rc = upload_real_field(c, im.data(), c->d_shi + (size_t)w * cnt, g.Nh)) return rc;
        return check_launch(c, "update_fields");
    }
    if (int rc = upload_real_field(c, ch, c->d_ch + (size_t)w * g.Lt * g.Nh, g.Nh)) return rc;
    if (int rc = upload_real_field(c, sh, c->d_sh + (size_t)w * g.Lt * g.Nh, g.Nh)) return rc;
    if (!c->cs_const.empty()) {  // Lτ×Nh column-major: τ-independent when every column is constant
        bool same = true;
        for (int h = 0; h < g.Nh && same; ++h)
            for (int l = 1; l < g.Lt && same; ++l) same = ch[(size_t)h * g.Lt + l] == ch[(size_t)h * g.Lt] && sh[(size_t)h * g.Lt + l] == sh[(size_t)h * g.Lt];
        set_cs_const(c, w, !same ? 0 : std::min(cs_level_of(c, ch, (size_t)g.Lt), cs_level_of(c, sh, (size_t)g.Lt)));
    }
    launch_pack_csf(c->stream, c->d_ch + (size_t)w * g.Lt * g.Nh, c->d_sh + (size_t)w * g.Lt * g.Nh, c->d_psrc, c->d_csf + (size_t)w * g.Lt * c->kg.ptotal, c->d_cs_varies + w, g.Lt, g.Lt, g.Nh, c->kg.ptotal);
    return check_launch(c, "update_fields");
}

// update!(fdm, fpi) for walkers [w0, w0 + nw): arrays of the walkers are stacked back to back.
// V == NULL or t == NULL leaves that part of the fields as it is.
static int update_pi_range(smoqy_ctx *c, int w0, int nw, const double *V, const double *t, const int64_t *perm, double dtau)
{
    const Geometry &g = c->g;
    const size_t nV = (size_t)g.Lt * g.N, nT = (size_t)g.Lt * g.Nh;
    const size_t tw = g.is_cplx ? 2 : 1;  // T = ComplexF64: t is complex128
    if (int rc = ensure_stage_real(c, (size_t)nw * (nV + tw * nT) + 2)) return rc;
    if (int rc = ensure_stage_int(c, (size_t)std::max(g.Nh, 1))) return rc;
    double *dV = c->d_stage_real, *dT = c->d_stage_real + (size_t)nw * nV + ((size_t)nw * nV & 1);  // 16-byte aligned for double2
    if (V) HIPCHK(c, hipMemcpyAsync(dV, V, (size_t)nw * nV * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (t && nT) {
        if (!perm) FAIL(c, 1, "perm must be given with t");
        std::vector<int> p0((size_t)g.Nh);
        for (int h = 0; h < g.Nh; ++h) {
            if (perm[h] < 1 || perm[h] > g.Nh) FAIL(c, 1, "perm[%d] = %lld out of range", h + 1, (long long)perm[h]);
            p0[h] = (int)perm[h] - 1;
        }
        HIPCHK(c, hipMemcpyAsync(dT, t, (size_t)nw * nT * tw * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (int rc = pin_h2d(c, c->d_stage_int, p0.data(), p0.size() * sizeof(int))) return rc;  // p0 is a temporary: through the page-locked arena
    }
    const bool do_t = t && nT;
    if (g.is_cplx) {
        launch_fields_from_path_integral_c(c->stream, V ? dV : nullptr, do_t ? (const double2 *)dT : nullptr, c->d_stage_int, c->d_expV + (size_t)w0 * nV, c->d_ch + (size_t)w0 * nT, c->d_sh + (size_t)w0 * nT,
                                           c->d_shi + (size_t)w0 * nT, nw * g.Lt, g.N, g.Nh, dtau, g.is_sym ? dtau / 2 : dtau);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return check_launch(c, "update_from_path_integral");
    }
    launch_fields_from_path_integral(c->stream, V ? dV : nullptr, do_t ? dT : nullptr, c->d_stage_int, c->d_expV + (size_t)w0 * nV, c->d_ch + (size_t)w0 * nT, c->d_sh + (size_t)w0 * nT, nw * g.Lt, g.N, g.Nh,
                                     dtau, g.is_sym ? dtau / 2 : dtau);  // FermionDetMatrix.jl:220
    if (do_t) {
        launch_pack_csf(c->stream, c->d_ch + (size_t)w0 * nT, c->d_sh + (size_t)w0 * nT, c->d_psrc, c->d_csf + (size_t)w0 * g.Lt * c->kg.ptotal, c->d_cs_varies + w0, nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal);
        for (int w = 0; w < nw && !c->cs_const.empty(); ++w) {  // t is Nh×Lτ column-major per walker: equal hoppings on every slice give equal cosh / sinh
            const double *tw_ = t + (size_t)w * nT;
            bool same = true;
            for (int l = 1; l < g.Lt && same; ++l)
                for (int h = 0; h < g.Nh && same; ++h) same = tw_[(size_t)l * g.Nh + h] == tw_[h];
            int level = same ? 1 : 0;
            if (same) {  // ... and equal hoppings on every bond of a colour (sorted bond n is model hopping perm[n]) give one pair per colour
                std::vector<double> ts((size_t)g.Nh);
                for (int n = 0; n < g.Nh; ++n) ts[(size_t)n] = tw_[perm[n] - 1];
                level = cs_level_of(c, ts.data(), 1);
            }
            set_cs_const(c, w0 + w, level);
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "update_from_path_integral");
}

int smoqy_update_from_path_integral(smoqy_ctx *c, int w, const double *V, const double *t, const int64_t *perm, double dtau)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    HIPCHK(c, hipSetDevice(c->device));
    return update_pi_range(c, w, 1, V, t, perm, dtau);
}

int smoqy_update_from_path_integral_all(smoqy_ctx *c, const double *V_all, const double *t_all, const int64_t *perm, double dtau)
{
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    return update_pi_range(c, 0, c->g.nw, V_all, t_all, perm, dtau);
}

int smoqy_get_fields(smoqy_ctx *c, int w, double *expV, double *ch, double *sh)
{
    CHECK_CTX(c);
    CHECK_WALKER(c, w);
    const Geometry &g = c->g;
    HIPCHK(c, hipSetDevice(c->device));
    if (expV) if (int rc = download_real_field(c, c->d_expV + (size_t)w * g.Lt * g.N, expV, g.N)) return rc;
    if (g.is_cplx) {  // complex128 out, as the reference's Matrix{T} fields
        const size_t cnt = (size_t)g.Lt * g.Nh;
        std::vector<double> re(cnt), im(cnt);
        if (ch) {
            if (int rc = download_real_field(c, c->d_ch + (size_t)w * cnt, re.data(), g.Nh)) return rc;
            for (size_t k = 0; k < cnt; ++k) { ch[2 * k] = re[k]; ch[2 * k + 1] = 0.0; }
        }
        if (sh) {
            if (int rc = download_real_field(c, c->d_sh + (size_t)w * cnt, re.data(), g.Nh)) return rc;
            if (int rc = download_real_field(c, c->d_shi + (size_t)w * cnt, im.data(), g.Nh)) return rc;
            for (size_t k = 0; k < cnt; ++k) { sh[2 * k] = re[k]; sh[2 * k + 1] = im[k]; }
        }
        return check_launch(c, "get_fields");
    }
    if (ch) if (int rc = download_real_field(c, c->d_ch + (size_t)w * g.Lt * g.Nh, ch, g.Nh)) return rc;
    if (sh) if (int rc = download_real_field(c, c->d_sh + (size_t)w * g.Lt * g.Nh, sh, g.Nh)) return rc;
    return check_launch(c, "get_fields");
}

// ---- vectors ------------------------------------------------------------------------------------

int smoqy_vec_alloc(smoqy_ctx *c, int *id)
{
    CHECK_CTX(c);
    HIPCHK(c, hipSetDevice(c->device));
    double2 *p = nullptr;
    HIPCHK(c, hipMalloc(&p, c->vec_elems() * sizeof(double2)));
    HIPCHK(c, hipMemsetAsync(p, 0, c->vec_elems() * sizeof(double2), c->stream));
    for (size_t k = 0; k < c->vecs.size(); ++k)
        if (!c->vecs[k]) { c->vecs[k] = p; *id = (int)k; return 0; }
    c->vecs.push_back(p);
    *id = (int)c->vecs.size() - 1;
    return 0;
}

int smoqy_vec_free(smoqy_ctx *c, int id)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(c->vecs[id]));
    c->vecs[id] = nullptr;
    return 0;
}

#define CHECK_RANGE(c, sys0, count) \
    if ((sys0) < 0 || (count) < 1 || (sys0) + (count) > (c)->g.nsys) FAIL(c, 1, "system range [%d, %d) outside 0..%d", (sys0), (sys0) + (count), (c)->g.nsys)

static int upload_into(smoqy_ctx *c, double2 *dev, const void *host, int sys0, int count)
{
    const Geometry &g = c->g;
    const size_t bytes = (size_t)count * g.Lt * g.N * sizeof(double2);
    HIPCHK(c, hipMemcpyAsync(c->d_stage, host, bytes, hipMemcpyHostToDevice, c->stream));
    launch_transpose_in(c->stream, c->d_stage, dev, g.Lt, g.N, g.nsys, sys0, count);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "upload");
}

static int download_from(smoqy_ctx *c, const double2 *dev, void *host, int sys0, int count)
{
    const Geometry &g = c->g;
    const size_t bytes = (size_t)count * g.Lt * g.N * sizeof(double2);
    launch_transpose_out(c->stream, dev, c->d_stage, g.Lt, g.N, g.nsys, sys0, count);
    HIPCHK(c, hipMemcpyAsync(host, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "download");
}

int smoqy_vec_upload(smoqy_ctx *c, int id, const void *host, int sys0, int count)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    return upload_into(c, c->vecs[id], host, sys0, count);
}

int smoqy_vec_download(smoqy_ctx *c, int id, void *host, int sys0, int count)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, id)) return rc;
    CHECK_RANGE(c, sys0, count);
    HIPCHK(c, hipSetDevice(c->device));
    return download_from(c, c->vecs[id], host, sys0, count);
}

int smoqy_vec_copy(smoqy_ctx *c, int dst, int src)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, dst)) return rc;
    if (int rc = check_vec(c, src)) return rc;
    if (dst != src) HIPCHK(c, hipMemcpyAsync(c->vecs[dst], c->vecs[src], c->vec_elems() * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int smoqy_vec_dot(smoqy_ctx *c, int a, int b, void *out)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, a)) return rc;
    if (int rc = check_vec(c, b)) return rc;
    const Geometry &g = c->g;
    launch_dot(c->stream, c->vecs[a], c->vecs[b], c->part_c, c->d_dot_out, g.Lt, g.N, g.nsys, c->Tc, c->nchunk);
    if (int rc = pin_d2h(c, out, c->d_dot_out, (size_t)g.nsys * sizeof(double2))) return rc;
    return check_launch(c, "vec_dot");
}

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
        R = c->Tc;
        while (2 * R <= 32 && (long)((g.Lt + 2 * R - 1) / (2 * R)) * count >= 1024) R *= 2;
    }
    R = std::min(R, g.Lt);
    R -= R % c->Tc;
    return std::max(R, 0);
}

static int matvec_dev(smoqy_ctx *c, int op, double2 *out, const double2 *in, double2 *partial, const CgState *cg, int sys0, int count, bool twiddled = false, hipStream_t st = nullptr)
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
        name = c->fw.kind == 1 ? "fdm_wave_kernel<ring>" : (c->fw.kind == 2 ? "fdm_wave_kernel<plaquette>" : "fdm_wave_kernel<honeycomb block>");
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
static int pstat_wait(smoqy_ctx *c);
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
static int pstat_wait(smoqy_ctx *c)
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
static int precond_update_range(smoqy_ctx *c, int w0, int nw, const double *randvecs, const double *d_randvecs = nullptr)
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
    c->ff.enabled = (!on && !g.is_cplx && g.ncol >= 1 && g.ncol <= kFdmColours && c->ff.threads <= 1024) ? 1 : 0;
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
static int precond_core(smoqy_ctx *c, const double2 *src, double2 *v, const CgState *cg, double2 *part_rz, bool half = false)
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

// form of the handle's own τ-FFT: 0 = two LDS images (fewer passes: fastest alone), 1 = in place (fewer registers and half the LDS: more
// workgroups per CU, better when several handles share the GPU)
int smoqy_tfft_form(smoqy_ctx *c, int in_place)
{
    CHECK_CTX(c);
    if (in_place != 0 && in_place != 1) FAIL(c, 1, "in_place must be 0 or 1");
    const int want = (in_place && c->tf_ok && c->tf.slim_ok && c->tf.pos) ? 1 : 0;  // lengths with a factor 7 keep the two-image form
    if (want != c->tf.slim) {
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
static int set_part_streams(smoqy_ctx *c, int nparts)
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

static int auto_parts(const smoqy_ctx *c) { return c->g.nsys >= 8 ? 2 : 1; }  // measured, DESIGN.md §4.3

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
static int cg_dev(smoqy_ctx *c, double2 *x, const double2 *b, bool x_is_b, double tol, int maxiter, int use_precond, int *iters, double *eps, const double2 *pff_phi = nullptr,
                  double2 *pff_out = nullptr, int async_step = -1)
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

static ForceArgs force_args(smoqy_ctx *c, double nu, const double2 *u, const double2 *v)
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
    launch_dmdx(c->stream, a, c->g.is_sym != 0);
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
    launch_dmdx(c->stream, a, g.is_sym != 0);                                                                        // -2 Re<AΨ|∂M/∂x|ΛΨ>   :150
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
        launch_pack_csf(c->stream, c->d_ch, c->d_sh, c->d_psrc, c->d_csf, c->d_cs_varies, g.nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal);
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
static int pff_core(smoqy_ctx *c, int phi, int psi, const double *randvec_all, double tol, int maxiter, int use_precond, bool want_force, int *iters, double *eps,
                    const double *d_randvec_all = nullptr, double2 *d_dot = nullptr, int async_step = -1)
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

#define CHECK_EFA(c)                                                                                                  \
    if (!(c)->force.set || !(c)->force.bare_set) FAIL(c, 1, "call smoqy_force_set_couplings and smoqy_set_bare_model first"); \
    if (!(c)->force.efa_set) FAIL(c, 1, "call smoqy_efa_config first")

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
        launch_pack_csf(c->stream, c->d_ch, c->d_sh, c->d_psrc, c->d_csf, c->d_cs_varies, g.nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal);
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

// ---- measurement aids -------------------------------------------------------------------------------

int smoqy_timer_start(smoqy_ctx *c)
{
    CHECK_CTX(c);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return 0;
}

int smoqy_timer_stop(smoqy_ctx *c, double *ms)
{
    CHECK_CTX(c);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    return 0;
}

int smoqy_matvec_timing(smoqy_ctx *c, int sample_every, int max_samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sample_every < 0 || max_samples < 0 || max_samples > 65536) FAIL(c, 1, "invalid sampling parameters");
    while ((int)T.ev.size() < max_samples) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(c, hipEventCreate(&a));
        HIPCHK(c, hipEventCreate(&b));
        T.ev.push_back({a, b});
    }
    {   // one (start, end) slot per workgroup per sampled launch, at most 1024 launches (16 MiB at 1024 workgroups)
        const int cap = std::min(max_samples, 1024), wgs = c->g.Lt * c->g.nsys;  // nchunk <= Lt
        if (cap > T.stamp_cap || wgs != T.stamp_wgs) {
            if (T.d_stamp) (void)hipFree(T.d_stamp);
            T.d_stamp = nullptr;
            T.stamp_cap = T.stamp_wgs = 0;
            if (cap > 0) {
                HIPCHK(c, hipMalloc(&T.d_stamp, 2 * (size_t)cap * wgs * sizeof(unsigned long long)));
                T.stamp_cap = cap;
                T.stamp_wgs = wgs;
            }
        }
        if (T.d_stamp) HIPCHK(c, hipMemset(T.d_stamp, 0, 2 * (size_t)T.stamp_cap * T.stamp_wgs * sizeof(unsigned long long)));
    }
    T.every = sample_every;
    T.seen = T.used = 0;
    return 0;
}

// the same sampled launches by the device's own clock: mean of (last workgroup's end - first workgroup's start), the interval
// rocprofv3 --kernel-trace reports for a dispatch.  Call BEFORE smoqy_matvec_timing_read (which ends the sampling).
int smoqy_matvec_timing_read_device(smoqy_ctx *c, double *avg_us, int *samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = std::min(T.used, T.stamp_cap);
    const size_t per = 2 * (size_t)T.stamp_wgs;
    std::vector<unsigned long long> st(per * (size_t)std::max(n, 1));
    if (n) HIPCHK(c, hipMemcpy(st.data(), T.d_stamp, per * n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0.0;
    int cnt = 0;
    for (int k = 0; k < n; ++k) {
        unsigned long long t0 = ~0ull, t1 = 0ull;
        for (int b = 0; b < T.stamp_wgs; ++b) {
            const unsigned long long s0 = st[per * k + 2 * b], s1 = st[per * k + 2 * b + 1];
            if (s0) t0 = std::min(t0, s0);   // slots of workgroups that never ran (smaller grid) stay 0
            t1 = std::max(t1, std::max(s0, s1));  // a workgroup that retired at entry (converged system) only has a start
        }
        if (t0 != ~0ull && t1 > t0) { sum += (double)(t1 - t0) * 0.01; ++cnt; }  // 100 MHz ticks -> µs
    }
    if (avg_us) *avg_us = cnt ? sum / cnt : 0.0;
    if (samples) *samples = cnt;
    return 0;
}

int smoqy_matvec_timing_read(smoqy_ctx *c, double *avg_us, int *samples)
{
    CHECK_CTX(c);
    auto &T = c->mvt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (int k = 0; k < T.used; ++k) {
        float f = 0;
        HIPCHK(c, hipEventElapsedTime(&f, T.ev[k].first, T.ev[k].second));
        sum += f;
    }
    if (avg_us) *avg_us = T.used ? 1e3 * sum / T.used : 0.0;
    if (samples) *samples = T.used;
    T.every = 0;
    T.seen = T.used = 0;
    return 0;
}

// per-kernel durations of the fused CG iteration inside real solves: the next `iterations` full-batch iterations on the handle's stream get an
// event in front of each of their four launches (MᵀM, forward τ-FFT, Chebyshev, inverse τ-FFT) and one behind the last
int smoqy_cg_iteration_timing(smoqy_ctx *c, int iterations)
{
    CHECK_CTX(c);
    if (iterations < 0 || iterations > 4096) FAIL(c, 1, "invalid number of iterations");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto &IT = c->itt;
    while (IT.ev.size() < (size_t)5 * iterations) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreate(&e));
        IT.ev.push_back(e);
    }
    IT.want = iterations;
    IT.used = 0;
    return 0;
}

// us[0..3] = mean event-to-event time of the four launches over the sampled iterations (dependent launches on one stream: the kernel plus
// the hand-over to the next one); ends the sampling
int smoqy_cg_iteration_timing_read(smoqy_ctx *c, double *us, int *iterations)
{
    CHECK_CTX(c);
    if (!us) FAIL(c, 1, "us is NULL");
    auto &IT = c->itt;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double sum[4] = {0, 0, 0, 0};
    for (int k = 0; k < IT.used; ++k)
        for (int q = 0; q < 4; ++q) {
            float f = 0;
            HIPCHK(c, hipEventElapsedTime(&f, IT.ev[(size_t)5 * k + q], IT.ev[(size_t)5 * k + q + 1]));
            sum[q] += f;
        }
    for (int q = 0; q < 4; ++q) us[q] = IT.used ? 1e3 * sum[q] / IT.used : 0.0;
    if (iterations) *iterations = IT.used;
    IT.want = IT.used = 0;
    return 0;
}

int smoqy_bench_matvec(smoqy_ctx *c, int op, int out, int in, int reps, double *ms)
{
    CHECK_CTX(c);
    if (int rc = check_vec(c, out)) return rc;
    if (int rc = check_vec(c, in)) return rc;
    if (out == in) FAIL(c, 1, "bench_matvec needs distinct vectors");
    if (int rc = matvec_dev(c, op, c->vecs[out], c->vecs[in], nullptr, nullptr, 0, c->g.nsys)) return rc;  // warm-up
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r)
        if (int rc = matvec_dev(c, op, c->vecs[out], c->vecs[in], nullptr, nullptr, 0, c->g.nsys)) return rc;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    return check_launch(c, "bench_matvec");
}

// device stream-copy ceiling: `reps` copies of `bytes` bytes (src -> dst, both allocated here, far larger than the 256 MiB
// Infinity Cache when bytes >= 1 GiB) between two HIP events on the handle's stream; moved bytes = 2 * bytes per copy
int smoqy_bench_copy(smoqy_ctx *c, size_t bytes, int reps, double *ms)
{
    CHECK_CTX(c);
    if (bytes < 16 || reps < 1) FAIL(c, 1, "invalid copy benchmark parameters");
    const size_t n = bytes / sizeof(double2);
    double2 *src = nullptr, *dst = nullptr;
    HIPCHK(c, hipMalloc(&src, n * sizeof(double2)));
    if (hipMalloc(&dst, n * sizeof(double2)) != hipSuccess) { (void)hipFree(src); FAIL(c, 2, "out of device memory for the copy benchmark"); }
    int rc = 0;
    float f = 0;
    do {
        if (hipMemsetAsync(src, 1, n * sizeof(double2), c->stream) != hipSuccess) { rc = 2; break; }
        launch_stream_copy(c->stream, dst, src, n);  // warm-up (page faults, clocks)
        if (hipEventRecord(c->ev0, c->stream) != hipSuccess) { rc = 2; break; }
        for (int r = 0; r < reps; ++r) launch_stream_copy(c->stream, dst, src, n);
        if (hipEventRecord(c->ev1, c->stream) != hipSuccess || hipEventSynchronize(c->ev1) != hipSuccess) { rc = 2; break; }
        if (hipEventElapsedTime(&f, c->ev0, c->ev1) != hipSuccess) { rc = 2; break; }
    } while (0);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (rc) FAIL(c, rc, "HIP error in the copy benchmark: %s", hipGetErrorString(hipGetLastError()));
    *ms = f;
    return check_launch(c, "bench_copy");
}

int smoqy_algorithmic_bytes(const smoqy_ctx *c, int op, double *bytes)
{
    CHECK_CTX(c);
    const Geometry &g = c->g;
    const double V = (double)g.Lt * g.N;
    const double S = 16.0 * V, F = 8.0 * V + 16.0 * g.Lt * g.Nh;
    const double one = g.nsys * 2.0 * S + g.nw * F;
    *bytes = (op == SMOQY_OP_MTM || op == SMOQY_OP_MMT) ? 2.0 * one : one;
    return 0;
}

}  // extern "C"
