// Private header of the C ABI translation units (api_*.hip): the handle (smoqy_ctx), the error macros and the internal functions one
// unit defines and another calls.  Nothing here is exported: csrc/smoqy.map keeps every symbol but smoqy_* local to libsmoqy_hip.so.
#pragma once
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
    double *d_pcsi = nullptr;  // T = ComplexF64, Sym: τ-mean of Im sinh per padded bond [nw][ptotal] beside d_pcs (cheb_fast_kernel<…, CPLX>)
    double *d_csi = nullptr;   // T = ComplexF64, Sym: Im sinh per padded bond [nw][Lt][ptotal] beside d_csf (fdm_fast_kernel<…, CPLX>)
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
    int traj_margin = 4;   // early-exit iterations launched past the hint: 4 x ~10 us per solve against a repeated trajectory per miss (3 of 93 at margin 2)
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
    int tf_rb_plan = 0;       // tfft_plan's register-blocked R·M·R choice (tf.rb is that, or 0 where the rule of tfft_rb_rule switches it off)
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

#define CHECK_WALKER(c, w) \
    if ((w) < 0 || (w) >= (c)->g.nw) FAIL(c, 1, "walker %d out of range 0..%d", (w), (c)->g.nw - 1)

#define CHECK_RANGE(c, sys0, count) \
    if ((sys0) < 0 || (count) < 1 || (sys0) + (count) > (c)->g.nsys) FAIL(c, 1, "system range [%d, %d) outside 0..%d", (sys0), (sys0) + (count), (c)->g.nsys)

#define CHECK_EFA(c)                                                                                                  \
    if (!(c)->force.set || !(c)->force.bare_set) FAIL(c, 1, "call smoqy_force_set_couplings and smoqy_set_bare_model first"); \
    if (!(c)->force.efa_set) FAIL(c, 1, "call smoqy_efa_config first")

// ---- internal functions shared between the api_*.hip units (defined where the comment says) ----
extern std::once_flag g_rocfft_once;  // api_handle.hip
extern "C" {
void drop_graphs(smoqy_ctx *c);  // api_handle.hip
void set_cs_const(smoqy_ctx *c, int w, int level);  // api_handle.hip
int cs_level_of(const smoqy_ctx *c, const double *v, size_t stride);  // api_handle.hip
int check_vec(smoqy_ctx *c, int id);  // api_handle.hip
int check_launch(smoqy_ctx *c, const char *what);  // api_handle.hip
void choose_chunking(smoqy_ctx *c);  // api_handle.hip
FdmArgs fdm_args(smoqy_ctx *c, const double2 *in, double2 *out, double2 *partial, const CgState *cg, int sys0, int count);  // api_handle.hip
KpmArgs kpm_args(smoqy_ctx *c, double2 *v, const CgState *cg);  // api_handle.hip
int ensure_stage_real(smoqy_ctx *c, size_t n);  // api_handle.hip
int ensure_stage_int(smoqy_ctx *c, size_t n);  // api_handle.hip
int pin_reserve(smoqy_ctx *c, size_t bytes, char **slot);  // api_handle.hip
int pin_h2d(smoqy_ctx *c, void *dst, const void *src, size_t bytes);  // api_handle.hip
int pin_d2h(smoqy_ctx *c, void *dst, const void *src, size_t bytes);  // api_handle.hip
void ge_release(smoqy_ctx *c);  // api_handle.hip
int set_part_streams(smoqy_ctx *c, int nparts);  // api_cg.hip
int auto_parts(const smoqy_ctx *c);  // api_cg.hip
int coef_table_stride(const smoqy_ctx *c);  // api_handle.hip
int upload_real_field(smoqy_ctx *c, const double *host, double *dev, int n);  // api_handle.hip
int download_real_field(smoqy_ctx *c, const double *dev, double *host, int n);  // api_handle.hip
int upload_into(smoqy_ctx *c, double2 *dev, const void *host, int sys0, int count);  // api_handle.hip
int download_from(smoqy_ctx *c, const double2 *dev, void *host, int sys0, int count);  // api_handle.hip
int matvec_dev(smoqy_ctx *c, int op, double2 *out, const double2 *in, double2 *partial, const CgState *cg, int sys0, int count, bool twiddled = false, hipStream_t st = nullptr);  // api_operator.hip
int pstat_wait(smoqy_ctx *c);  // api_precond.hip
void tfft_rb_rule(smoqy_ctx *c, bool shared_gpu);  // api_cg.hip: where tfft_rb_kernel runs (shared_gpu: the caller asked for the in-place form, smoqy_tfft_form)
int precond_update_range(smoqy_ctx *c, int w0, int nw, const double *randvecs, const double *d_randvecs = nullptr);  // api_precond.hip
int precond_core(smoqy_ctx *c, const double2 *src, double2 *v, const CgState *cg, double2 *part_rz, bool half = false);  // api_precond.hip
int cg_dev(smoqy_ctx *c, double2 *x, const double2 *b, bool x_is_b, double tol, int maxiter, int use_precond, int *iters, double *eps, const double2 *pff_phi = nullptr, double2 *pff_out = nullptr, int async_step = -1);  // api_cg.hip
ForceArgs force_args(smoqy_ctx *c, double nu, const double2 *u, const double2 *v);  // api_force.hip
int pff_core(smoqy_ctx *c, int phi, int psi, const double *randvec_all, double tol, int maxiter, int use_precond, bool want_force, int *iters, double *eps, const double *d_randvec_all = nullptr, double2 *d_dot = nullptr, int async_step = -1);  // api_force.hip
}
