// C ABI of libsmoqy_hip.so (include/smoqy_hip.h), part "handle": handle lifetime, layout conversion at the boundary, fields, vectors.
// gfx950 / ROCm only; there is no CPU path.  Split out of one api.hip in round 4; the handle and the shared internals are in ctx.h.
#include "ctx.h"

namespace {
std::string g_create_error;
}  // namespace
std::once_flag g_rocfft_once;  // rocfft_setup() once per process (this unit and api_greens.hip)

// destroy every captured CG iteration: called whenever something baked into the captured kernel arguments changes
void drop_graphs(smoqy_ctx *c)
{
    c->graph_epoch++;
    for (auto &gph : c->graphs) {
        if (gph.exec) { (void)hipGraphExecDestroy(gph.exec); gph.exec = nullptr; }
        if (gph.graph) { (void)hipGraphDestroy(gph.graph); gph.graph = nullptr; }
    }
}

// host-side proof that walker w's hoppings do (not) depend on τ; a change drops the captured CG graphs, which hold the kernel variant
// level: 0 unknown / τ-dependent, 1 τ-independent, 2 τ-independent AND the same (cosh, sinh) on every bond of a colour
void set_cs_const(smoqy_ctx *c, int w, int level)
{
    if (c->cs_const.empty()) return;
    if (c->cs_const[(size_t)w] != (char)level) {
        c->cs_const[(size_t)w] = (char)level;
        drop_graphs(c);
    }
}
// 2 when v[h] is the same for all sorted bonds h of each colour, else 1 (v: one value per sorted bond, a τ-independent hopping table)
int cs_level_of(const smoqy_ctx *c, const double *v, size_t stride)
{
    const Geometry &g = c->g;
    for (int col = 0; col < g.ncol; ++col) {
        const int h0 = (int)c->in_cr[2 * (size_t)col] - 1, h1 = (int)c->in_cr[2 * (size_t)col + 1];  // 1-based inclusive range
        for (int h = h0 + 1; h < h1; ++h)
            if (v[(size_t)h * stride] != v[(size_t)h0 * stride]) return 1;
    }
    return 2;
}

int check_vec(smoqy_ctx *c, int id)
{
    if (id < 0 || id >= (int)c->vecs.size() || !c->vecs[id]) FAIL(c, 1, "invalid vector id %d", id);
    return 0;
}

int check_launch(smoqy_ctx *c, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) FAIL(c, 2, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
    return 0;
}

void choose_chunking(smoqy_ctx *c)
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

FdmArgs fdm_args(smoqy_ctx *c, const double2 *in, double2 *out, double2 *partial, const CgState *cg, int sys0, int count)
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

KpmArgs kpm_args(smoqy_ctx *c, double2 *v, const CgState *cg)
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

int ensure_stage_real(smoqy_ctx *c, size_t n)
{
    if (n <= c->stage_real_cap) return 0;
    if (c->d_stage_real) (void)hipFree(c->d_stage_real);
    c->d_stage_real = nullptr;
    HIPCHK(c, hipMalloc(&c->d_stage_real, n * sizeof(double)));
    c->stage_real_cap = n;
    return 0;
}

int ensure_stage_int(smoqy_ctx *c, size_t n)
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
int pin_reserve(smoqy_ctx *c, size_t bytes, char **slot)
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

int pin_h2d(smoqy_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    char *slot = nullptr;
    if (int rc = pin_reserve(c, bytes, &slot)) return rc;
    std::memcpy(slot, src, bytes);
    HIPCHK(c, hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// small device -> host transfer into caller memory: lands in the arena, the stream is synchronised, then a plain memcpy
int pin_d2h(smoqy_ctx *c, void *dst, const void *src, size_t bytes)
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

void ge_release(smoqy_ctx *c)
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
                    c->d_rand, c->d_rand_traj, c->d_traj_dot, c->d_lan, c->d_order, c->d_active, c->d_coefs, c->d_pbonds, c->d_poff, c->d_psrc, c->d_pcs, c->d_csf, c->d_cs_varies, c->d_psites, c->d_pos, c->d_own, c->d_own_f == c->d_own ? nullptr : c->d_own_f, c->d_wave, c->d_fwave, c->d_big, c->d_shi, c->d_sbari, c->d_csi, c->d_pcsi};
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


// Stride of the per-slot coefficient table: the largest expansion order an ACTIVE preconditioner can ask for.  order = ⌊(ϵmax − ϵmin)(a1/ϕ + a2)⌋
// (KPMPreconditioner.jl:711) with 0 < ϵmin < 1 < ϵmax < 2 (:573) and ϕ ≥ π/Lτ (:220, folded :710), so order < 2 (a1 Lτ/π + a2).  Sizing the
// table for it once means the device can accept new bounds without the host growing anything (at Lτ = 128: 165 entries per slot).
int coef_table_stride(const smoqy_ctx *c)
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
        c->tf_rb_plan = c->tf.rb;
        tfft_rb_rule(c, false);
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
        if (g.is_cplx && g.is_sym && g.ncol >= 1 && g.ncol <= kMaxColours && maxp <= 1024) {  // complex hoppings: the Sym Chebyshev kernel on the padded lists (round 4)
            HIPCHK(c, hipMalloc(&c->d_pcsi, (size_t)g.nw * std::max<size_t>(pb.size(), 1) * sizeof(double)));
            HIPCHK(c, hipMemset(c->d_pcsi, 0, (size_t)g.nw * std::max<size_t>(pb.size(), 1) * sizeof(double)));
            c->kg.pcsi = c->d_pcsi;
            c->kg.cplx_fast = 1;
        }
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
        if (g.is_cplx && g.is_sym) {
            HIPCHK(c, hipMalloc(&c->d_csi, (size_t)g.nw * g.Lt * std::max<size_t>(pb.size(), 1) * sizeof(double)));
            HIPCHK(c, hipMemset(c->d_csi, 0, (size_t)g.nw * g.Lt * std::max<size_t>(pb.size(), 1) * sizeof(double)));
        }
        HIPCHK(c, hipMalloc(&c->d_cs_varies, (size_t)g.nw * sizeof(int)));
        HIPCHK(c, hipMemset(c->d_cs_varies, 0, (size_t)g.nw * sizeof(int)));
        c->ff.cs_varies = c->d_cs_varies;
        c->cs_const.assign((size_t)g.nw, 0);
        c->ff.psites = c->d_psites; c->ff.pos = c->d_pos;
        c->ff.pbonds = c->d_pbonds; c->ff.poff = c->d_poff; c->ff.csf = c->d_csf; c->ff.csi = c->d_csi; c->ff.ptotal = c->kg.ptotal; c->ff.threads = c->kg.threads;
        // Sym: fdm_fast / own / stream / wave kernels; Asym: fdm_fast_asym_kernel; complex hoppings: Sym only, fdm_fast_kernel<…, CPLX> (the other families test ff.csi)
        c->ff.enabled = ((!g.is_cplx || g.is_sym) && g.ncol >= 1 && g.ncol <= kFdmColours && maxp <= 1024) ? 1 : 0;
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
             !c->tf_ok ? "rocFFT + cg_update kernels"
             : c->tf.edge ? "tfft_kernel (register-blocked 4*M*4 form)"
             : c->tf.rb   ? "tfft_rb_kernel (register-blocked R*M*R form)"
                          : (c->tf.slim ? "tfft_kernel (in-place form)" : "tfft_kernel (two-image form)"));
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

int upload_real_field(smoqy_ctx *c, const double *host, double *dev, int n)
{
    const size_t cnt = (size_t)c->g.Lt * n;
    if (cnt == 0) return 0;
    if (int rc = ensure_stage_real(c, cnt)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_stage_real, host, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_transpose_real_in(c->stream, c->d_stage_real, dev, c->g.Lt, n);
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is reused by the next field
    return 0;
}

int download_real_field(smoqy_ctx *c, const double *dev, double *host, int n)
{
    const size_t cnt = (size_t)c->g.Lt * n;
    if (cnt == 0) return 0;
    if (int rc = ensure_stage_real(c, cnt)) return rc;
    launch_transpose_real_out(c->stream, dev, c->d_stage_real, c->g.Lt, n);
    HIPCHK(c, hipMemcpyAsync(host, c->d_stage_real, cnt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}


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
        if (c->d_csi)
            launch_pack_csf(c->stream, c->d_ch + (size_t)w * cnt, c->d_sh + (size_t)w * cnt, c->d_psrc, c->d_csf + (size_t)w * g.Lt * c->kg.ptotal, c->d_cs_varies + w, g.Lt, g.Lt, g.Nh, c->kg.ptotal,
                            c->d_shi + (size_t)w * cnt, c->d_csi + (size_t)w * g.Lt * c->kg.ptotal);
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
        if (do_t && c->d_csi)
            launch_pack_csf(c->stream, c->d_ch + (size_t)w0 * nT, c->d_sh + (size_t)w0 * nT, c->d_psrc, c->d_csf + (size_t)w0 * g.Lt * c->kg.ptotal, c->d_cs_varies + w0, nw * g.Lt, g.Lt, g.Nh, c->kg.ptotal,
                            c->d_shi + (size_t)w0 * nT, c->d_csi + (size_t)w0 * g.Lt * c->kg.ptotal);
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


int upload_into(smoqy_ctx *c, double2 *dev, const void *host, int sys0, int count)
{
    const Geometry &g = c->g;
    const size_t bytes = (size_t)count * g.Lt * g.N * sizeof(double2);
    HIPCHK(c, hipMemcpyAsync(c->d_stage, host, bytes, hipMemcpyHostToDevice, c->stream));
    launch_transpose_in(c->stream, c->d_stage, dev, g.Lt, g.N, g.nsys, sys0, count);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "upload");
}

int download_from(smoqy_ctx *c, const double2 *dev, void *host, int sys0, int count)
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


}  // extern "C"
