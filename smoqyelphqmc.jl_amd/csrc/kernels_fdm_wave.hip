// Fused MᵀM with ONE WAVEFRONT PER RUN OF SLICES and the whole time slice in registers (round 4).
//
// Reference: mul_MtM! = mul_Mt!(mul_M!) (src/FermionDetMatrix.jl:329-340, 385-427, 484-525) with checkerboard_lmul!
// (src/checkerboard_matrix_multiply.jl:50-69) for the Sym propagator B_l = C_{L-1} … C_1 (C_0 D_l C_0) C_1 … C_{L-1}.
//
// fdm_stream_kernel walks a run of slices with a WORKGROUP: the slice lives in LDS, every colour stage is a read-modify-write of
// LDS with a workgroup barrier behind it (five barriers per output slice at three colours, nine at four), and with that stage chain
// compiled out the same kernel runs at the HBM specification ("the stage chain is half of the kernel at every size").
// Here the stage chain has no LDS image and no barrier at all: the sites of a slice are dealt out to the 64 lanes of ONE wavefront
// in groups chosen so that most bonds join two registers of the same lane and the others reach a fixed neighbouring lane:
//
//   ring       2 colours, 4 sites per lane: r[4l … 4l+3] of the cycle the two colours form        (bond-SSH chain, N <= 256)
//   plaquette  4 colours, 4 sites per lane: a 4-cycle of colours 1 and 2                            (optical-SSH square lattice, N <= 256)
//   block      3 colours, 8 sites per lane: 2 x 2 unit cells (A, B) of the honeycomb lattice       (Holstein honeycomb, N <= 512)
//
// A colour stage is then, per own site p, one multiply-add pair with a partner value that is either another register of the lane
// (no instruction) or a register of another lane, passed through a wavefront-private LDS image (DPP wave rotations on a ring of 64 lanes).  The centre
// stage C_0 D C_0 is folded into two coefficients per site (as in the Chebyshev kernels).  A wavefront walks `run_len` output slices:
//
//     y[m]     = v[m] − h·B_m v[m−1]          m = la … lb          (rows of M; the last one is halo recompute)
//     out[m−1] = y[m−1] − h̄·B_m y[m]          m = la+1 … lb        (rows of Mᵀ)
//
// with the next slice and its fields requested one iteration ahead, straight from global memory into registers; `out` is stored from
// registers; p·Ap is Σ|y[m]|² over the run's own slices, as in fdm_stream_kernel.  No s_barrier in the kernel; LDS only as the wavefront's
// private exchange image for values that live in another lane (rings of 64 lanes: DPP wave rotations, no LDS at all).
// The host finds the groups from the neighbour table alone (fdm_wave_program below, called from api_handle.hip) and verifies every relation the kernel relies on;
// lattices that do not fit keep fdm_stream_kernel / fdm_fast_kernel.  Per-site arithmetic equals theirs up to the folded centre
// stage (two roundings instead of five) — compared with the oracle at 1e-13 like every operator kernel.
#include "smoqy_internal.h"
#include "wave_desc.h"

#include <algorithm>
#include <array>
#include <type_traits>
#include <utility>
#include <vector>

namespace smoqy {
using namespace wave_desc;
namespace {

__device__ __forceinline__ int wrapl(int l, int Lt) { return l >= Lt ? l - Lt : (l < 0 ? l + Lt : l); }
__device__ __forceinline__ double2 lin(double a, double2 x, double b, double2 y) { return make_double2(a * x.x + b * y.x, a * x.y + b * y.y); }
__device__ __forceinline__ double2 hopcomb(double2 v, double2 u, bool wrap, bool dagger, const FdmArgs &a)
{
    double pr = a.hop_re, pi = dagger ? -a.hop_im : a.hop_im;
    if (wrap && a.antiperiodic) { pr = -pr; pi = -pi; }
    return make_double2(v.x - (pr * u.x - pi * u.y), v.y - (pr * u.y + pi * u.x));
}

template <int K>
__device__ __forceinline__ void opaque(double (&d)[K])
{
#pragma unroll
    for (int i = 0; i < K; ++i) asm volatile("" : "+v"(d[i]));
}
__device__ __forceinline__ double2 shfl2(double2 x, int lane)
{
    return make_double2(__shfl(x.x, lane, 64), __shfl(x.y, lane, 64));
}
// the fields of one time slice as a lane needs them: (cosh, sinh) per bond slot (CSM = 2: τ-dependent, reloaded per slice), the two
// folded centre coefficients per own site
template <class D, int CSM>
struct SliceFields {
    double2 cs[CSM == 2 ? D::NB : 1];
    double e0[D::S], e1[D::S];
    double d[D::S];   // LEAN form: exp(-ΔτV) itself, the two centre coefficients are formed where they are used
};

// CSM: 0 — one (cosh, sinh) per COLOUR for the whole launch (the host has shown the hoppings τ-independent and uniform per colour:
// the Holstein models), kept in scalar registers; 1 — one per bond slot, τ-independent; 2 — one per bond slot and slice (SSH models)
// MINW: wavefronts per SIMD the register allocation must leave room for (the second __launch_bounds__ argument).  1 everywhere but in the
// honeycomb-block twin below: at MINW = 1 that program takes 256 VGPRs + 2 AGPRs = 264 allocated registers, i.e. ONE wavefront per SIMD
// (tools/kernel_resources.py; 512 // 264); at MINW = 2 the LEAN form below (exp(-ΔτV) kept instead of the two folded centre coefficients per
// site, which are formed again inside each propagate: 16 more fp64 operations per slice) takes 252 registers, no scratch, no AGPR.
template <class D, int CSM, bool ROT, int MINW = 1>
__global__ void __launch_bounds__(64, MINW) fdm_wave_kernel(FdmArgs a, FdmFast ff, FdmWave fw)
{
    constexpr int S = D::S, NCOL = D::NCOL, NB = D::NB, NR = D::NR;
    __shared__ double2 xch[ROT ? 1 : S * 64];
    const int Lt = a.Lt, N = a.N, lane = threadIdx.x;
    const int R = a.run_len, nrun = (Lt + R - 1) / R;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    if ((nblk & 7) == 0) bid = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);  // XCD-aware order: consecutive runs of a system share halo slices
    const int run = bid % nrun, sys = a.sys_first + bid / nrun;
    stamp_begin(a.stamp);
    const int sys_done = a.cg[sys].done;
    const int w = sys / a.nrhs;
    const int la = run * R, lb = min(Lt, la + R);
    const size_t sstride = (size_t)a.nsys * N;
    const double2 *in = a.in + (size_t)sys * N;
    double2 *out = a.out + (size_t)sys * N;
    const double *expV = a.expV + (size_t)w * Lt * N;
    const double2 *csf = ff.csf + (size_t)w * Lt * ff.ptotal;

    // ---- the lane's program: site ids, bond slots, partner lanes ----
    const bool on = lane < fw.lanes;
    const int lq = on ? lane : 0;  // lanes past the groups run group 0's program on zeros and never store
    int site[S], msite[D::REMOTE0 ? S : 1], bnd[CSM ? NB : 1], rl[NR];
    static_for<0, S>([&](auto P) { site[P] = fw.tab[P * 64 + lq]; });
    if constexpr (CSM != 0) static_for<0, NB>([&](auto B) { bnd[B] = fw.tab[(S + B) * 64 + lq]; });
    static_for<0, NR>([&](auto Rw) { rl[Rw] = on ? fw.tab[(S + NB + Rw) * 64 + lq] : lane; });
    if constexpr (D::REMOTE0) static_for<0, S>([&](auto P) { msite[P] = fw.tab[(S + NB + NR + P) * 64 + lq]; });
    (void)msite; (void)bnd;

    // (cosh, sinh) that do not change along the run
    double2 ucs[CSM == 0 ? NCOL : 1], kcs[CSM == 1 ? NB : 1];
    if constexpr (CSM == 0) static_for<0, NCOL>([&](auto C) { ucs[C] = csf[ff.poff[C]]; });   // slice 0, first bond of the colour: the same for all
    if constexpr (CSM == 1) static_for<0, NB>([&](auto B) { kcs[B] = csf[bnd[B]]; });
    (void)ucs; (void)kcs;

    auto load_slice = [&](double2 (&x)[S], int m) {
        const double2 *row = in + (size_t)wrapl(m, Lt) * sstride;
        static_for<0, S>([&](auto P) { x[P] = row[site[P]]; });
    };
    // raw fields of slice m: exp(-ΔτV) at the own sites (and at the colour-0 partners where those live in other lanes), (c, s) per slot
    struct Raw { double d[S], dm[D::REMOTE0 ? S : 1]; };
    auto load_raw = [&](Raw &r, int m) {
        const int l = wrapl(m, Lt);
        const double *ev = expV + (size_t)l * N;
        static_for<0, S>([&](auto P) {
            r.d[P] = a.nt_fields ? __builtin_nontemporal_load(&ev[site[P]]) : ev[site[P]];
            if constexpr (D::REMOTE0) r.dm[P] = ev[msite[P]];
        });
    };
    SliceFields<D, CSM> F;
    // τ-dependent (cosh, sinh): ONE register set, refilled for slice m + 1 as soon as the second propagate of slice m has read it (a
    // second set would cost 4·NB registers and, on the plaquette program, the second wavefront per SIMD that hides this very load)
    auto load_cs = [&](int m) {
        if constexpr (CSM == 2) {
            const int l = wrapl(m, Lt);
            static_for<0, NB>([&](auto B) { F.cs[B] = csf[(size_t)l * ff.ptotal + bnd[B]]; });
        }
    };
    auto CS = [&](auto C, auto P) -> double2 {
        if constexpr (CSM == 0) return ucs[C];
        else if constexpr (CSM == 1) return kcs[D::bs(decltype(C)::value, decltype(P)::value)];
        else return F.cs[D::bs(decltype(C)::value, decltype(P)::value)];
    };
    constexpr bool LEAN = MINW >= 2 && !D::REMOTE0;
    auto set_fields = [&](const Raw &r) {
        if constexpr (LEAN) {
            static_for<0, S>([&](auto P) { F.d[P] = r.d[P]; });
            return;
        }
        static_for<0, S>([&](auto P) {
            // C₀ D C₀ on own site p with partner q: x' = (c² d_p + s² d_q) x_p + c s (d_p + d_q) x_q
            const double2 k = CS(std::integral_constant<int, 0>{}, P);
            double dq;
            if constexpr (D::REMOTE0) dq = r.dm[P]; else dq = r.d[D::pp(0, decltype(P)::value)];
            F.e0[P] = k.x * k.x * r.d[P] + k.y * k.y * dq;
            F.e1[P] = k.x * k.y * (r.d[P] + dq);
        });
    };
    // partner values of colour C for every own position
    // Values that live in another lane travel through the wavefront's own LDS image [position][lane] (16-byte writes and reads: half the
    // LDS-pipe time of four ds_bpermute_b32 per value, which is what bounded the first form of this kernel — SQ_ACTIVE_INST_LDS was a
    // third of the launch at 16 systems).  No barrier: the image belongs to ONE wavefront and the LDS queue of a wavefront is in order.
    auto partners = [&](auto C, const double2 (&x)[S], double2 (&m)[S]) {
        if constexpr (!ROT) {
            static_for<0, S>([&](auto P) {
                if constexpr (D::rr(decltype(C)::value, decltype(P)::value) >= 0) xch[decltype(P)::value * 64 + lane] = x[P];
            });
        }
        static_for<0, S>([&](auto P) {
            constexpr int q = D::pp(decltype(C)::value, decltype(P)::value), r = D::rr(decltype(C)::value, decltype(P)::value);
            if constexpr (r < 0) m[P] = x[q];
            else if constexpr (ROT) m[P] = make_double2(wave_rot<(r == 0 ? 0x134 : 0x13C)>(x[q].x), wave_rot<(r == 0 ? 0x134 : 0x13C)>(x[q].y));
            else m[P] = xch[q * 64 + rl[r]];
        });
    };
    auto stage = [&](auto C, double2 (&x)[S]) {
        double2 m[S];
        partners(C, x, m);
        static_for<0, S>([&](auto P) {
            const double2 k = CS(C, P);
            x[P] = lin(k.x, x[P], k.y, m[P]);
        });
    };
    // x <- B x with the fields in F
    auto apply_B = [&](double2 (&x)[S]) {
        static_for<1, NCOL>([&](auto I) { stage(std::integral_constant<int, NCOL - decltype(I)::value>{}, x); });     // C_{L-1} … C_1
        {
            double2 m[S];
            partners(std::integral_constant<int, 0>{}, x, m);
            if constexpr (LEAN) {
                // the coefficients of C₀ D C₀ from exp(-ΔτV) on the spot (the same expressions as set_fields): 8 registers instead of 24-32 across
                // both propagates of a slice; the empty asm keeps the compiler from carrying them from one propagate to the next
                opaque(F.d);
                static_for<0, S>([&](auto P) {
                    const double2 k = CS(std::integral_constant<int, 0>{}, P);
                    const double dp = F.d[P], dq = F.d[D::pp(0, decltype(P)::value)];
                    const double e0 = k.x * k.x * dp + k.y * k.y * dq;
                    const double e1 = k.x * k.y * (dp + dq);
                    x[P] = lin(e0, x[P], e1, m[P]);
                });
            } else
            static_for<0, S>([&](auto P) { x[P] = lin(F.e0[P], x[P], F.e1[P], m[P]); });           // C_0 D C_0
        }
        static_for<1, NCOL>([&](auto I) { stage(I, x); });                                            // C_1 … C_{L-1}
    };

    // ---- walk the run ----
    double2 vprev[S], vcur[S], vnext[S], y[S], yprev[S];
    Raw raw;
    load_slice(vprev, la - 1);
    load_slice(vcur, la);
    load_raw(raw, la);
    load_cs(la);
    asm volatile("" ::: "memory");  // keeps the loads above on this side of the early return (no instruction, no wait)
    if (sys_done) return;           // wavefront-uniform; nothing has been stored yet
    double accr = 0.0;
    for (int m = la; m <= lb; ++m) {
        set_fields(raw);
        if (m < lb) {  // the next slice and its exp(-ΔτV), one iteration ahead
            load_slice(vnext, m + 1);
            load_raw(raw, m + 1);
        }
        // y[m] = v[m] − h B_m v[m−1]
        apply_B(vprev);
        const bool wrap_m = wrapl(m, Lt) == 0;
        static_for<0, S>([&](auto P) { y[P] = hopcomb(vcur[P], vprev[P], wrap_m, false, a); });
        if (m < lb && on) static_for<0, S>([&](auto P) { accr += y[P].x * y[P].x + y[P].y * y[P].y; });  // |y[m]|², m a slice of this run
        if (m > la) {
            // out[m−1] = y[m−1] − h̄ B_m y[m]
            double2 u[S];
            static_for<0, S>([&](auto P) { u[P] = y[P]; });
            apply_B(u);
            const bool wrap_o = (m - 1) == Lt - 1;
            double2 *row = out + (size_t)(m - 1) * sstride;
            if (m < lb) load_cs(m + 1);
            if (on) static_for<0, S>([&](auto P) { row[site[P]] = hopcomb(yprev[P], u[P], wrap_o, true, a); });
        } else if (m < lb) {
            load_cs(m + 1);
        }
        static_for<0, S>([&](auto P) { yprev[P] = y[P]; vprev[P] = vcur[P]; vcur[P] = vnext[P]; });
    }
    if (a.partial) {
        for (int off = 32; off > 0; off >>= 1) accr += __shfl_down(accr, off, 64);
        if (lane == 0) {
            // the consumers reduce a.nchunk partials per system: the run's sum goes to its first chunk, its other chunks are zero
            const int c0 = la / a.Tc, c1 = (lb + a.Tc - 1) / a.Tc;
            a.partial[(size_t)sys * a.nchunk + c0] = make_double2(accr, 0.0);
            for (int c = c0 + 1; c < c1; ++c) a.partial[(size_t)sys * a.nchunk + c] = make_double2(0.0, 0.0);
        }
    }
    stamp_end(a.stamp);
}

template <class D, bool ROT>
void launch_kind(hipStream_t st, const FdmArgs &a, const FdmFast &ff, const FdmWave &fw, int csm)
{
    const int nrun = (a.Lt + a.run_len - 1) / a.run_len;
    const dim3 grid((unsigned)(nrun * a.sys_count)), block(64);
    if (csm == 0) hipLaunchKernelGGL((fdm_wave_kernel<D, 0, ROT>), grid, block, 0, st, a, ff, fw);
    else if (csm == 1) hipLaunchKernelGGL((fdm_wave_kernel<D, 1, ROT>), grid, block, 0, st, a, ff, fw);
    else hipLaunchKernelGGL((fdm_wave_kernel<D, 2, ROT>), grid, block, 0, st, a, ff, fw);
}

}  // namespace

// csm: 0 uniform per colour and τ-independent, 1 τ-independent, 2 τ-dependent hoppings (what the host has shown for every walker of the launch)
bool fdm_wave_supported(const FdmArgs &a, const FdmFast &ff, const FdmWave &fw, bool sym, int csm)
{
    static const int env = tuning_env(kTuneFdmWave);
    if (env == 0 || !sym || !ff.enabled || ff.csi || fw.kind == 0 || !fw.tab || a.run_len < 1 || a.run_len % a.Tc != 0 || a.Lt < 2) return false;
    if (fw.kind == 3 && csm != 0) return false;  // eight sites per lane leave no registers for a (cosh, sinh) pair per bond slot
    return true;
}

void launch_fdm_wave(hipStream_t st, const FdmArgs &a, const FdmFast &ff, const FdmWave &fw, int csm)
{
    static const int env = tuning_env(kTuneFdmWave);
    switch (fw.kind) {
        case 1:
            if (fw.rot && env != 2) launch_kind<RingD, true>(st, a, ff, fw, csm);
            else launch_kind<RingD, false>(st, a, ff, fw, csm);
            break;
        case 2: launch_kind<PlaqD, false>(st, a, ff, fw, csm); break;
        default: {
            // SMOQY_FDM_WAVE_OCC=2: the two-wavefronts-per-SIMD twin (same expressions, 252 registers).  Built at the end of round 4 from the
            // static register counts, NOT yet timed on a GPU: off unless asked for (wave_run_length aims at 2048 wavefronts with it).
            static const int occ = tuning_env(kTuneFdmWaveOcc);
            // (launched directly, not through launch_kind: the per-bond hopping modes of this program are refused by fdm_wave_supported and
            // would only add two never-launched 314- and 380-register kernels to the library)
            const int nrun = (a.Lt + a.run_len - 1) / a.run_len;
            const dim3 grid((unsigned)(nrun * a.sys_count)), block(64);
            if (occ == 2) hipLaunchKernelGGL((fdm_wave_kernel<HoneyD, 0, false, 2>), grid, block, 0, st, a, ff, fw);
            else hipLaunchKernelGGL((fdm_wave_kernel<HoneyD, 0, false>), grid, block, 0, st, a, ff, fw);
            break;
        }
    }
}


// ---- host: find the groups (lane programs) of a decomposition ---------------------------------------------------------------------
namespace {
struct GroupDesc {
    int S, NCOL, NB, NR, kind;
    bool remote0;
    int pp[4][8], rr[4][8], bs[4][8];
};
template <class D>
GroupDesc make_desc()
{
    GroupDesc d{};
    d.S = D::S; d.NCOL = D::NCOL; d.NB = D::NB; d.NR = D::NR; d.kind = D::KIND; d.remote0 = D::REMOTE0;
    for (int c = 0; c < D::NCOL; ++c)
        for (int p = 0; p < D::S; ++p) { d.pp[c][p] = D::pp(c, p); d.rr[c][p] = D::rr(c, p); d.bs[c][p] = D::bs(c, p); }
    return d;
}

// Deal the sites out to groups of d.S positions such that EVERY relation of the descriptor holds: internal bonds join two positions of a
// group, remote bonds join position p of a group with position pp(c, p) of the group in row rr(c, p) of the lane table.  The labelling is
// propagated from one site and then checked relation by relation; false = this lattice is not of this kind.
bool build_groups(const GroupDesc &d, int N, const std::vector<std::vector<int>> &mate, const std::vector<std::vector<int>> &bidx, int origin_pos, std::vector<int> &tab, int &lanes,
                  bool &rot)
{
    if (N % d.S != 0 || N / d.S > 64) return false;
    const int n = N / d.S;
    std::vector<int> grp((size_t)N, -1), pos((size_t)N, -1), queue;
    std::vector<std::array<int, 8>> groups;
    auto assign = [&](int g, int p, int s) {
        if (grp[s] != -1) return grp[s] == g && pos[s] == p;
        if (groups[(size_t)g][(size_t)p] != -1) return false;
        groups[(size_t)g][(size_t)p] = s; grp[s] = g; pos[s] = p;
        return true;
    };
    auto new_group = [&](int s, int p) {
        if ((int)groups.size() >= n) return false;
        const int g = (int)groups.size();
        std::array<int, 8> e;
        e.fill(-1);
        groups.push_back(e);
        if (!assign(g, p, s)) return false;
        for (int sweep = 0; sweep < d.S; ++sweep)      // close the group under its internal bonds
            for (int c = 0; c < d.NCOL; ++c)
                for (int q = 0; q < d.S; ++q)
                    if (d.rr[c][q] < 0 && groups[(size_t)g][(size_t)q] != -1 && !assign(g, d.pp[c][q], mate[c][groups[(size_t)g][(size_t)q]])) return false;
        for (int q = 0; q < d.S; ++q)
            if (groups[(size_t)g][(size_t)q] == -1) return false;
        queue.push_back(g);
        return true;
    };
    if (!new_group(0, origin_pos)) return false;
    for (size_t h = 0; h < queue.size(); ++h) {
        const int g = queue[h];
        for (int c = 0; c < d.NCOL; ++c)
            for (int p = 0; p < d.S; ++p) {
                if (d.rr[c][p] < 0) continue;
                const int t = mate[c][groups[(size_t)g][(size_t)p]];
                if (grp[t] == -1) { if (!new_group(t, d.pp[c][p])) return false; }
                else if (pos[t] != d.pp[c][p]) return false;
            }
    }
    if ((int)groups.size() != n) return false;
    for (int s = 0; s < N; ++s)
        if (grp[s] == -1) return false;
    // lane = rank of the group by its smallest site id: neighbouring lanes read neighbouring memory
    std::vector<int> order((size_t)n), rank((size_t)n);
    for (int g = 0; g < n; ++g) order[(size_t)g] = g;
    auto minsite = [&](int g) { int m = N; for (int q = 0; q < d.S; ++q) m = std::min(m, groups[(size_t)g][(size_t)q]); return m; };
    std::sort(order.begin(), order.end(), [&](int x, int y) { return minsite(x) < minsite(y); });
    for (int l = 0; l < n; ++l) rank[(size_t)order[(size_t)l]] = l;
    const int rows = d.S + d.NB + d.NR + (d.remote0 ? d.S : 0);
    tab.assign((size_t)rows * 64, 0);
    for (int l = 0; l < n; ++l) {
        const std::array<int, 8> &G = groups[(size_t)order[(size_t)l]];
        for (int p = 0; p < d.S; ++p) tab[(size_t)p * 64 + l] = G[(size_t)p];
        std::vector<int> slot((size_t)d.NB, -1);
        for (int c = 0; c < d.NCOL; ++c)
            for (int p = 0; p < d.S; ++p) {
                const int s = G[(size_t)p], t = mate[c][s], q = d.pp[c][p], r = d.rr[c][p];
                // every relation, checked: the partner sits at position q of this group (internal) or of another one (remote)
                if (pos[t] != q || mate[c][t] != s) return false;
                if (r < 0) { if (grp[t] != grp[s]) return false; }
                else tab[(size_t)(d.S + d.NB + r) * 64 + l] = rank[(size_t)grp[t]];
                const int b = d.bs[c][p];
                if (slot[(size_t)b] == -1) slot[(size_t)b] = bidx[c][s];
                else if (slot[(size_t)b] != bidx[c][s]) return false;   // the two ends of an internal bond name the same padded bond
                if (bidx[c][t] != bidx[c][s]) return false;
                if (c == 0 && d.remote0) tab[(size_t)(d.S + d.NB + d.NR + p) * 64 + l] = t;
            }
        for (int b = 0; b < d.NB; ++b) {
            if (slot[(size_t)b] == -1) return false;
            tab[(size_t)(d.S + b) * 64 + l] = slot[(size_t)b];
        }
    }
    // a remote relation and its mirror must agree on the lane pair: if lane l reads position q of lane l' under colour c, then lane l' reads
    // position p of lane l under the same colour (the bond is one bond) — implied by mate[c][t] == s above.
    rot = false;
    if (d.kind == 1 && n == 64) {  // ring of exactly one wavefront: row 0 = the next lane, row 1 = the previous one -> DPP wave rotations
        rot = true;
        for (int l = 0; l < n && rot; ++l) rot = tab[(size_t)(d.S + d.NB + 0) * 64 + l] == (l + 1) % n && tab[(size_t)(d.S + d.NB + 1) * 64 + l] == (l + n - 1) % n;
    }
    lanes = n;
    return true;
}
}  // namespace

// mate[c][site], bidx[c][site]: partner site and padded-bond index of `site` under colour c (every colour a perfect matching: the caller
// has checked that).  kind = 0: no lane program, the handle keeps the workgroup kernels.
void fdm_wave_program(int N, int ncol, const std::vector<std::vector<int>> &mate, const std::vector<std::vector<int>> &bidx, std::vector<int> &tab, int &kind, int &lanes, bool &rot)
{
    kind = 0; lanes = 0; rot = false;
    const GroupDesc descs[3] = {make_desc<RingD>(), make_desc<PlaqD>(), make_desc<HoneyD>()};
    for (const GroupDesc &d : descs) {
        if (d.NCOL != ncol) continue;
        for (int origin = 0; origin < d.S; ++origin)   // site 0 may sit at any position of its group
            if (build_groups(d, N, mate, bidx, origin, tab, lanes, rot)) { kind = d.kind; return; }
    }
    tab.clear();
}

}  // namespace smoqy
