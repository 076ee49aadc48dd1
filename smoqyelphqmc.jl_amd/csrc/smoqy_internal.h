// Internal declarations shared by the translation units of libsmoqy_hip.so (gfx950 only).
//
// Device data layout ("slice-major", chosen for MI355X — see DESIGN.md §3):
//   state vector batch   v[l][s][i]   complex128, l = imaginary-time slice, s = system, i = site
//   per-walker fields    expV[w][l][i], cosh/sinh[w][l][h], Lambda[w][l][i]   float64
// i.e. one time slice of one system is a contiguous N-vector (perfectly coalesced for the
// checkerboard kernels and for the per-frequency Chebyshev recurrence), and the tau-FFT is a
// strided batched transform with stride nsys*N and distance 1.
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/smoqy_hip.h"

namespace smoqy {

constexpr int kThreads = 256;       // workgroup size of the slice kernels (4 wavefronts)
constexpr int kMaxPartials = 1024;  // upper bound on tau-chunks per system
constexpr int kMaxColours = 6;      // colours held in registers by the fast KPM kernels
constexpr int kFdmColours = 4;      // colours held in registers by the fast FermionDetMatrix kernels

struct Geometry {
    int Lt, N, Nh, ncol, nw, nrhs, nsys;
    int is_sym;
    int is_cplx;  // matrix-element type T = ComplexF64 (complex hoppings): generic kernels only
};

// per-system CG state living on the device
struct CgState {
    double normb2;      // |b|^2
    double rho_re, rho_im;  // r.z of the current iteration
    double alpha_re, alpha_im;  // α of the current iteration (the x update happens one kernel later than the r update)
    double eps;         // last relative residual
    int iters;          // completed iterations
    int done;           // 0 running, 1 converged, 2 maxiter reached; written by the LAST kernel of an iteration only
    int stop;           // `done` as it stood when the iteration began: latched by the FIRST update kernel of the next iteration
                        // (forward tau-FFT / cg_update_xr) and by cg_start.  The closing kernel of an iteration (inverse tau-FFT /
                        // cg_update_p) gates on `stop`, never on `done`, because one of its own workgroups writes `done` and a
                        // launch carries no co-residency guarantee: a workgroup dispatched after that write must still apply the
                        // final x += αp.
    int precond_on;     // this system's walker has an active preconditioner
    int maxiter;        // stop criteria live here (not in kernel arguments) so a captured iteration can be replayed
    double tol;
};

struct FdmArgs {
    int Lt, N, Nh, ncol, nsys, nrhs, Tc, nchunk;
    const int2 *bonds;      // [Nh] 0-based site pairs, colour sorted
    const int *col_off;     // [ncol+1] bond offsets of the colours
    const double *expV, *ch, *sh;
    const double *shi;      // imaginary part of sinhΔτt for T = ComplexF64 (same layout as sh), nullptr for real hoppings: the bond
                            // factor is [[c, s], [conj(s), c]] (src/checkerboard_matrix_multiply.jl:60-68); generic kernels only
    const double2 *in;
    double2 *out;
    double2 *partial;       // [nsys][nchunk] dot(in, out) partials, or nullptr
    const CgState *cg;      // optional: skip systems with done != 0
    int sys_first, sys_count;  // systems [sys_first, sys_first + sys_count) are processed
    double hop_re, hop_im;     // phase on the inter-slice hop: 1 for the reference operator, exp(-iπ/Lτ) inside the CG
    int antiperiodic;          // 1: the wrap-around row carries the opposite sign (reference operator)
    // lattices whose slices do not fit in LDS: the generic kernels stage their slices in this global scratch instead
    // (scratch_stride elements per workgroup, L2-resident; same code, same barriers)
    double2 *scratch;
    size_t scratch_stride;
    // measurement aid (smoqy_matvec_timing): when non-null, workgroup b of the launch stores the device's constant 100 MHz clock
    // into stamp[2b] at entry and stamp[2b + 1] at exit (plain stores to its own slot — a shared atomic counter would serialise
    // the 1024 workgroups at ~12 ns each and lengthen the launch it measures); the host takes min(start) / max(end): first start
    // -> last end of the launch, the interval rocprofv3 reports, free of the inter-launch gap an event pair includes
    unsigned long long *stamp;
    int run_len;  // fdm_stream_kernel: output slices per workgroup (a multiple of Tc); 0 = not a streaming launch
    int nt_fields;  // fdm_stream_kernel: nontemporal loads of exp(-ΔτV)
};

// stamps for FdmArgs::stamp
__device__ __forceinline__ void stamp_begin(unsigned long long *st)
{
    if (st && threadIdx.x == 0) st[2 * (size_t)blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void stamp_end(unsigned long long *st)
{
    if (st && threadIdx.x == 0) st[2 * (size_t)blockIdx.x + 1] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
}

struct KpmArgs {
    int Lt, N, Nh, ncol, nsys, nrhs, is_sym;
    const int2 *bonds;
    const int *col_off;
    const double *dbar, *cbar, *sbar;   // [w][N], [w][Nh], [w][Nh] tau-means
    const double *sbari;                // [w][Nh] tau-mean of Im sinhΔτt (T = ComplexF64), nullptr for real hoppings
    const int *order;                   // [w][nslot]
    const double2 *coefs;               // [w][nslot][maxorder]
    const double *bounds;               // [w][2]
    const int *active;                  // [w]
    int nslot, maxorder;
    int xcd_map;                        // cheb_own_kernel: keep a system's workgroups on one XCD (see tfft_kernel)
    double2 *v;                         // input, slice(=frequency)-major
    double2 *vout;                      // output; nullptr = in place
    double2 *scratch;                   // see FdmArgs::scratch
    size_t scratch_stride;
    const CgState *cg;
    double2 *part_rz;                   // optional [nsys][rz_stride]: Parseval partial of r·z per (system, ω)
    int rz_stride;
    double scale;                       // output scale (1/Lτ: rocFFT's inverse is unnormalised)
    int sys_first, sys_count;           // systems [sys_first, sys_first + sys_count) are processed; sys_count = 0 means all
    int half;                           // real-vector ldiv! (KPMPreconditioner.jl:312 / :444): only ω < cld(Lτ, 2) are evaluated,
                                        // launch_conj_mirror fills in the rest
    int heavy, group;                   // cheb_own_kernel: the `heavy` lowest-|ϕ| frequencies get a workgroup each, the others go `group` at a time
                                        // (0, 0 = defaults of launch_cheb)
};

// Device-resident bookkeeping of update_preconditioner! (src/KPMPreconditioner.jl:565-597): the Lanczos kernel ends with the tridiagonal
// extremes, the (1 ∓ rbuf) widening, the activation test and the "bounds moved by more than rbuf/2" decision; when the bounds are
// accepted it also fills the expansion orders (:696-731) and raises `rebuild`, which the expansion kernel (:734-795) acts on.  The host
// never waits for any of it: it reads `status` whenever it next synchronises with the stream for another reason.
struct PreUpd {
    double *bounds;   // [nw][2] the ACCEPTED bounds: what order / coefs were built for (Chebyshev kernels read these)
    int *active;      // [nw]
    int *order;       // [nw][nslot]
    double2 *coefs;   // [nw][nslot][maxorder]
    int *rebuild;     // [nw] 1: this update accepted new bounds, the expansion kernel must refill the coefficients
    int *status;      // [nw][4] = {number of rebuilds so far, 2·(last slot with order > 1) + 2 capped at Lτ (0: none), largest order, active}
    double rbuf, a1, a2;  // a1 already doubled for Sym (:263)
    int nslot, maxorder, Lt, is_sym;
};

// geometry of the KPM fast path: per-colour bond lists padded with identity self bonds (i, i) so
// that every colour covers all N sites; lane t of a workgroup owns padded bond poff[c] + t
struct KpmGeom {
    const int2 *pbonds;  // [ptotal] LDS positions of the two sites of each padded bond
    const int2 *psites;  // [ptotal] the site ids themselves (global-memory addressing)
    const int *pos;      // [N] site -> LDS position: first-colour x sites, then its y sites, so that on a
                         // bipartite lattice the lanes of a wavefront touch consecutive 16-byte slots (no bank conflicts)
    const int *poff;     // [ncol + 1] (device)
    const int *psrc;     // [ptotal] source bond index, -1 for a self bond
    double2 *pcs;        // [nw][ptotal] tau-averaged (cosh, sinh) per padded bond
    double *pcsi;        // [nw][ptotal] tau-averaged Im sinh per padded bond (T = ComplexF64, Sym), else nullptr
    int cplx_fast;       // complex hoppings, Sym, geometry within the padded lists: cheb_fast_kernel<true, NCOL, CPLX> although fast = 0
    int ptotal;
    int threads;         // workgroup size = padded bonds per colour rounded up to a wavefront
    int fast;            // 0: use the generic kernels
    // owner-computes layout of the Sym Chebyshev kernel: lane j keeps the two sites of the j-th padded bond of
    // colour own_q in registers for the whole chain.  own is [(4 + 4 ncol)][threads] ints: the lane's two site ids,
    // the site ids of their colour-0 mates, then per colour the LDS slots of the two mates and the padded-bond
    // indices (into pcs) of the two bonds.
    const int *own;
    int own_q, own_n;
    int wl0;             // the colour-0 mates of every lane's two sites are held by lanes of the same wavefront, first site's mate in a second
                         // slot and vice versa: the centre exchange of cheb_own_kernel runs on wave shuffles (own_chain<…, WL0>)
    // one-wavefront-per-chain lane program (kernels_kpm_wave.hip): 0 none, 1 ring (2 colours), 2 plaquette (4 colours); wave is
    // [rows][64] ints (rows = 11 / 28), wave_lanes <= 64 lanes own four sites each
    const int *wave;
    int wave_kind, wave_lanes;
};

// geometry + packed hopping table of the register-resident FermionDetMatrix kernels
struct FdmFast {
    const int2 *pbonds;  // padded bond lists as LDS positions (shared with KpmGeom)
    const int2 *psites;  // the same as site ids
    const int *pos;      // [N] site -> LDS position
    const int *poff;     // [ncol + 1] (device)
    const double2 *csf;  // [nw][Lt][ptotal] (cosh, sinh) per padded bond, (1, 0) on self bonds
    const double *csi;   // [nw][Lt][ptotal] Im sinh per padded bond for T = ComplexF64 (fdm_fast_kernel<…, CPLX>), nullptr for real hoppings
    const int *cs_varies; // [nw] 0 when a walker's hoppings are the same on every time slice
    int ptotal;
    int threads;
    int enabled;
    const int *own;      // owner-computes tables (layout as KpmGeom::own) for owned colour 1 (0 when there is one colour)
    int own_n;
    int full;            // every colour is a perfect matching: `threads` two-site bonds per padded list, N = 2·threads (fdm_stream_kernel<…, FULL>)
    int wl0;             // fdm_own_kernel: the colour-0 mates are the lane's neighbours in its row of 16 lanes (2 / 3 as KpmGeom::wl0, else 0)
};

// lane program of fdm_wave_kernel (kernels_fdm_wave.hip): one wavefront holds a whole time slice in registers
struct FdmWave {
    const int *tab;   // [rows][64]: site ids per position, padded-bond index per bond slot, partner lane per remote relation (+ colour-0 partner sites)
    int kind;         // 0 none, 1 ring (2 colours), 2 plaquette (4 colours), 3 honeycomb 2x2 cell blocks (3 colours)
    int lanes;        // groups = active lanes (<= 64)
    int rot;          // ring of exactly 64 lanes in lane order: partner fetches are DPP wave rotations
};

// ---- launchers (defined in the .hip files) -----------------------------------------------
void launch_fdm(hipStream_t st, int op, bool sym, const FdmArgs &a, size_t lds_bytes);
size_t fdm_lds_bytes(int op, int N, int Tc);
void launch_checkerboard(hipStream_t st, const FdmArgs &a, int inverse, int transposed, int col0, int ncols);
// each configure_* raises the dynamic-LDS limit of its kernels (once, outside any stream capture) and returns the first HIP
// error with the kernel's name in *what; smoqy_create fails on it instead of running kernels with a silently smaller limit
hipError_t configure_fdm_kernels(const char **what);
hipError_t configure_kpm_kernels(const char **what);
#define SMOQY_SET_LDS(fn, bytes)                                                                                                       \
    do {                                                                                                                               \
        hipError_t _e = hipFuncSetAttribute((const void *)(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));                   \
        if (_e != hipSuccess && first == hipSuccess) { first = _e; *what = #fn; }                                                       \
    } while (0)
bool fdm_own_stream_supported(const FdmArgs &a, const FdmFast &ff, bool sym, bool cs_const);
void launch_fdm_own_stream(hipStream_t st, const FdmArgs &a, const FdmFast &ff);
bool fdm_fast_supported(const FdmArgs &a, const FdmFast &ff, bool sym);
// cs_const: the caller has shown, on the host, that the hoppings of every walker of this launch do not depend on τ (Sym form only)
void launch_fdm_fast(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff, bool sym = true, bool cs_const = false);
// streaming form of the fused MᵀM (workgroups walk runs of a.run_len slices, loads two iterations ahead of the stage chain)
bool fdm_stream_supported(const FdmArgs &a, const FdmFast &ff, bool sym);
void launch_fdm_stream(hipStream_t st, const FdmArgs &a, const FdmFast &ff, bool cs_const);
hipError_t configure_fdm_stream_kernels(const char **what);
// fused MᵀM, one wavefront per run of slices (csm: 0 hoppings uniform per colour and τ-independent, 1 τ-independent, 2 τ-dependent)
bool fdm_wave_supported(const FdmArgs &a, const FdmFast &ff, const FdmWave &fw, bool sym, int csm);
void launch_fdm_wave(hipStream_t st, const FdmArgs &a, const FdmFast &ff, const FdmWave &fw, int csm);
void fdm_wave_program(int N, int ncol, const std::vector<std::vector<int>> &mate, const std::vector<std::vector<int>> &bidx, std::vector<int> &tab, int &kind, int &lanes, bool &rot);
bool fdm_own_supported(const FdmArgs &a, const FdmFast &ff, bool sym);
void launch_fdm_own(hipStream_t st, int op, const FdmArgs &a, const FdmFast &ff);
void launch_pack_csf(hipStream_t st, const double *ch, const double *sh, const int *psrc, double2 *csf, int *cs_varies, int Lt, int Lt1, int Nh, int ptotal, const double *shi = nullptr, double *csi = nullptr);

// device stream-copy ceiling (kernels_vec.hip): dst[i] = src[i], 16 bytes per lane, grid-stride
void launch_stream_copy(hipStream_t st, double2 *dst, const double2 *src, size_t n);
void launch_transpose_in(hipStream_t st, const double2 *host_layout, double2 *dev_layout, int Lt, int N, int nsys, int sys0, int count);
void launch_transpose_out(hipStream_t st, const double2 *dev_layout, double2 *host_layout, int Lt, int N, int nsys, int sys0, int count);
void launch_transpose_real_in(hipStream_t st, const double *src, double *dst, int Lt, int n);   // (Lt x n col-major) -> [l][n]
void launch_transpose_real_out(hipStream_t st, const double *src, double *dst, int Lt, int n);  // [l][n] -> (Lt x n col-major)
void launch_fields_from_path_integral(hipStream_t st, const double *V, const double *t, const int *perm0, double *expV, double *ch, double *sh, int Lt, int N, int Nh, double dtau, double dtau_k);
// T = ComplexF64: t is complex128 [l][h]; sinh carries sign(conj t) (src/FermionDetMatrix.jl:231)
void launch_fields_from_path_integral_c(hipStream_t st, const double *V, const double2 *t, const int *perm0, double *expV, double *ch, double *sh, double *shi, int Lt, int N, int Nh, double dtau, double dtau_k);
void launch_lambda_update(hipStream_t st, double *Lam, int Lt, int N, const double *x, int Nph, double dtau, int ncoup, const int *c2p, const int *c2s, const double *alpha, const double *alpha3, const int *phsym, const int *site_first, const int *site_next, int Lt1);
void launch_lambda_apply(hipStream_t st, int op, double2 *out, const double2 *in, const double *Lam, int Lt, int N, int nsys, int nrhs, int wslot_override);
void launch_dot(hipStream_t st, const double2 *a, const double2 *b, double2 *partial, double2 *out, int Lt, int N, int nsys, int Tc, int nchunk);
void launch_dot_final(hipStream_t st, const double2 *partial, double2 *out, int nsys, int nchunk);
void launch_fft_twiddle(hipStream_t st, double2 *v, const double2 *tw, int Lt, int N, int nsys, int inverse);
void launch_make_twiddle(hipStream_t st, double2 *tw, int Lt, double scale);
void launch_tau_means(hipStream_t st, const KpmGeom &kg, const double *expV, const double *ch, const double *sh, double *dbar, double *cbar, double *sbar, int Lt, int N, int Nh, int w0, int nw, const double *shi = nullptr,
                      double *sbari = nullptr);
// alpha/beta: [nw][1024] each; randvec: [nw][N] doubles, or [nw][N] complex128 when k.sbari != nullptr (T = ComplexF64)
// the kernel ends with the device-side bookkeeping of update_preconditioner! (PreUpd) for its walker
void launch_lanczos(hipStream_t st, const KpmArgs &k, const KpmGeom &kg, int w0, int nw, const double *randvec, int nsteps, double *alpha, double *beta, bool use_BtB, const PreUpd &u);
// update_kpm_expansion_coefs! (:734-795) on the device for walkers [w0, w0 + nw) whose `rebuild` flag is set
void launch_kpm_expansions(hipStream_t st, const PreUpd &u, int w0, int nw);
void launch_cheb(hipStream_t st, const KpmArgs &k, const KpmGeom &kg);
// one wavefront per chain (rings and plaquette lattices up to 256 sites, Sym, component split): kernels_kpm_wave.hip
bool cheb_wave_supported(const KpmArgs &k, const KpmGeom &kg);
const char *cheb_kernel_name(const KpmArgs &k, const KpmGeom &kg);  // the kernel launch_cheb picks for these arguments
void launch_cheb_wave(hipStream_t st, const KpmArgs &k, const KpmGeom &kg);
bool cheb_split_active(const KpmArgs &k, const KpmGeom &kg);  // true: the Chebyshev kernel writes 2·Lτ r·z partials per system instead of Lτ
// v[Lτ-1-ω] = conj(v[ω]) for ω < cld(Lτ, 2) (KPMPreconditioner.jl:334 / :468; the middle frequency of an odd Lτ conjugates itself)
void launch_conj_mirror(hipStream_t st, double2 *v, int Lt, int N, int nsys);
// boundary conversion of real vectors: host layout (Lτ x N x count doubles) <-> complex host-layout staging
void launch_real_to_complex(hipStream_t st, const double *re, double2 *z, size_t n);
void launch_complex_to_real(hipStream_t st, const double2 *z, double *re, size_t n);

// CG kernels (kernels_vec.hip): the loop runs in the twiddled basis, see there
struct CgArgs {
    int Lt, N, nsys, nrhs, Tc, nchunk;
    double2 *x, *r, *p, *z, *v;
    const double2 *th;            // [Lt] theta_l = exp(-i pi l / Lt)
    const double2 *b;
    double2 *part_pz;             // [nsys][nchunk]
    double2 *part_rz;             // [nsys][rz_stride]; nrz valid entries (Lt from the Chebyshev kernel, nchunk otherwise)
    double *part_rr, *part_bb;    // [nsys][nchunk]
    CgState *st;
    double tol;
    int maxiter;
    int use_precond;
    int nrz, rz_stride;
    // The solve of calculate_fermionic_action! (src/PFFCalculator.jl:97-109) with its Λ applies folded into the first and the last kernel
    // (round 3): when `lam` is set, cg_init forms the right-hand side b = Λ⁻ᵀΦ on the fly from Φ (`phi`; :97) instead of reading a prepared b,
    // and cg_finish writes Ψ = Λ⁻¹ x (:107) into `x_out` together with the per-chunk partials of Φ·Ψ (:109) into `part_dot` — the arithmetic
    // and the summation order of the three separate kernels they replace, seven vector passes fewer per solve.
    const double *lam;     // [w][Lt][N]
    const double2 *phi;
    double2 *x_out;
    double2 *part_dot;     // [nsys][nchunk]
};
void launch_cg_init(hipStream_t s, const CgArgs &a, bool x_is_b);
void launch_cg_start(hipStream_t s, const CgArgs &a);
void launch_cg_update_xr(hipStream_t s, const CgArgs &a);
void launch_cg_update_p(hipStream_t s, const CgArgs &a);
void launch_cg_finish(hipStream_t s, const CgArgs &a);

// force terms (kernels_force.hip)
struct ForceArgs {
    int Lt, N, Nh, ncol, nsys, nrhs, nw, Tc, nchunk;
    const int2 *bonds;
    const int *col_off;
    const double *expV, *ch, *sh, *lam;
    const double *shi;                  // Im sinhΔτt for T = ComplexF64 (bond factor [[c, s], [conj(s), c]]), nullptr for real hoppings
    const double2 *u, *v;
    double nu, dtau;
    int Nph, Nhol, Nssh, Q;             // Q = Nhol (dV) + 2 Nssh (dK, two passes) + Nhol (dΛ) contribution slots per (walker, slice)
    const double *x;                    // [nw][Lt][Nph]
    const int *h_c2p, *h_c2s, *h_phsym;
    const double *h_alpha, *h_alpha2, *h_alpha3, *h_alpha4;
    const int *s_c2p;                   // [Nssh][2]
    const double *s_alpha, *s_alpha2, *s_alpha3, *s_alpha4;
    const double *s_alpha_im, *s_alpha2_im, *s_alpha3_im, *s_alpha4_im;  // T = ComplexF64: imaginary parts of the SSH couplings (nullptr: real)
    const int *bond_ptr, *bond_cpl;     // CSR: checkerboard bond -> SSH couplings
    const int *ph_ptr, *ph_slot;        // CSR: phonon -> contribution slots
    const int *site_ptr, *site_cpl;     // CSR: site -> Holstein couplings, in coupling order
    const double *ph_sign;
    double *contrib;                    // [nw][Lt][Q]
    double2 *scratch;                   // see FdmArgs::scratch
    size_t scratch_stride;
    int cs_slice0;  // dmdx_kernel: every walker's hoppings are τ-independent (host-proved, smoqy_ctx::cs_const): the factors are read from slice 0
    int fac_lds;  // dmdx_kernel: the chunk's (cosh, sinh) are staged in LDS behind the two slice images (set by launch_dmdx)
};
hipError_t configure_force_kernels(const char **what);
void launch_dmdx(hipStream_t st, const ForceArgs &a, bool sym, const FdmFast *ff = nullptr);  // ff: the handle's padded-bond tables (dmdx_fast_kernel), or nullptr
void launch_dldx(hipStream_t st, const ForceArgs &a);
void launch_force_reduce(hipStream_t st, const ForceArgs &a, double *out);
// V(x), t(x) -> expV, cosh, sinh (+ Λ) for every walker from the device copy of the phonon fields
// t0s_im / shi: imaginary part of the bare hoppings / of sinhΔτt for T = ComplexF64 (nullptr for real hoppings)
void launch_phonon_fields(hipStream_t st, const ForceArgs &a, const double *V0, const double *t0s, double *expV, double *ch, double *sh, double *lam, double dtau_k, bool do_t,
                          const double *t0s_im = nullptr, double *shi = nullptr);

// GreensEstimator contractions (kernels_greens.hip)
void launch_ge_gather(hipStream_t st, const double2 *v, double2 *A, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int conj);
void launch_ge_product(hipStream_t st, const double2 *Ah, const double2 *Bh, double2 *P, size_t n2, int nrhs, int nw);
void launch_ge_slot_gather(hipStream_t st, const double2 *v, double2 *S, int Lt, int N, int nsys, int n_orb, int orb, int Nc, int L1, int L2, int r1, int r2, int conj);
void launch_ge_pair_product(hipStream_t st, const double2 *S0, const double2 *S1, const double2 *S2, const double2 *S3, double2 *X, double2 *Y, const int2 *pairs, int npairs, size_t n1, int second,
                            const double2 *tD, int conj_tD, const double2 *t0, int conj_t0);
void launch_ge_pair_reduce(hipStream_t st, const double2 *X, const double2 *Y, double2 *P, int npairs, size_t n1);
void launch_ge_finalize_pairs(hipStream_t st, const double2 *S, double2 *out, int Lt, int Nc, double scale);
void launch_ge_boundary(hipStream_t st, const double2 *gr, const double2 *r, double2 *part, double2 *out, int Lt, int N, int nsys, int nrhs, int n_orb, int og, int orr, int Nc, int L1, int L2, int sh1, int sh2,
                        const double2 *tD, int conj_tD, int ts1, int ts2, const double2 *t0, int conj_t0, int nslab, double scale);
void launch_ge_finalize_gd0(hipStream_t st, const double2 *S, double2 *out, int Lt, int Nc, int nw, double scale, int same_orbital);

// own tau-FFT (kernels_tfft.hip): Stockham passes over LDS site tiles, optionally fused with the CG updates
struct TfftArgs {
    int Lt, N, nsys, SB, ntile, nfac;
    int sys_first, sys_count;             // systems [sys_first, sys_first + sys_count) are processed; sys_count = 0 means all
    int x_stream;                         // inverse CG mode: nontemporal loads / stores for x
    int xcd_map;                          // blockIdx -> (tile, system) map that keeps a system's workgroups on one XCD
    int edge;                             // two-image form with its first and last radix-4 stage in registers: the radix M (4 or 8) of the one LDS pass, 0 = off
    int rb;                               // tfft_rb_kernel (Lτ = R·M·R on R·M·SB lanes): 16·R + M, 0 = off
    int fac[16];
    unsigned long long fpack, sfpack;     // fac / sfac packed four bits per radix: what the kernels read
    // in-place form (one LDS image): radix 2 / 3 / 4 / 5 factors `sfac` applied as decimation-in-frequency passes (forward) or, in reverse
    // order, decimation-in-time passes (inverse); element k of the spectrum sits at LDS row pos[k]
    int slim, slim_ok, snfac;             // slim: in-place form selected; slim_ok: Lt = 2^a 3^b 5^c, the in-place form exists
    int sfac[16];
    const int *pos;                       // [Lt]
    const double2 *wtab;                  // [Lt] exp(-2 pi i q / Lt)
    // plain modes (0 forward, 1 inverse): dst = FFT(pre_tw * src) * conj(post_tw)
    const double2 *src;
    double2 *dst;
    const double2 *pre_tw, *post_tw;
    // CG modes (2: x/r update + forward, 3: inverse + stop test + p update)
    double2 *x, *r, *p;
    const double2 *z;
    const double2 *part_rz, *part_pz;
    double *part_rr;
    int nrz, rz_stride, npz, pz_stride, nrr, rr_stride;
    CgState *st;
};
bool tfft_plan(int Lt, int N, TfftArgs &a);
void tfft_positions(const TfftArgs &a, int *pos);  // host: fills pos[0..Lt) for the in-place form

// Exact-Fourier-acceleration leapfrog on the phonon fields (call sites src/EFAPFFHMCUpdater.jl:142, 150, 202,
// 244 — SmoQyDQMC's ExactFourierAccelerator itself is not under /root/reference: parity unpinned).  x and p live as
// [nw][Lτ][Nph] doubles; per (frequency ω of the periodic τ transform, phonon mode) q is the eigenvalue of the harmonic bosonic action
// S_b = ½ Σ q |x̃|² and m the dynamical mass of the fictitious momenta, K = ½ Σ |p̃|²/m (unitary transform).
struct EfaArgs {
    int Lt, Nph, nw, SB, ntile, nfac;
    int fac[16];
    const double2 *wtab;
    double *x, *p;             // [nw][Lt][Nph]
    const double *force;       // optional [nw][Lt][Nph]: p -= kick * force before anything else (EFAPFFHMCUpdater.jl:196)
    double kick;
    const double *q, *m;       // [Lt][Nph]
    const int *finite_mass;    // [Nph] 0: infinite-mass mode, left untouched
    double dt;                 // evolve_eom! time (mode 0)
    int mode;                  // 0: kick + exact harmonic evolution by dt; 1: p <- F⁻¹ √m F p (momenta from unit normal deviates); 2: energies only
    double *part;              // [nw][ntile][2] partial (K, S_b) of the state on exit
};
void launch_efa(hipStream_t st, const EfaArgs &a);
hipError_t configure_tfft_kernels(const char **what);
void launch_tfft(hipStream_t st, int mode, const TfftArgs &a);


// ---- measurement switches ------------------------------------------------------------------------------------------
// Every environment variable the library looks at, in one place: A/B switches for the twins of the fast paths and a few tuning values,
// none needed for normal use.  Read once per process; tuning_env("NAME") is the variable's integer value, or -1 when it is not set.
enum TuningKnob {
    kTuneChebWl0, kTuneChebSplit, kTuneChebOwn, kTuneChebGroup, kTuneFdmStream, kTuneFdmOwn, kTuneFdmOwnMax, kTuneFdmOwnStream, kTuneNtFields,
    kTuneXStream, kTuneXcdMap, kTuneTfftSlim, kTuneTfftSb, kTuneTfftEdge, kTuneChebWave, kTuneFdmWave, kTuneFdmWaveR, kTuneDmdxFast, kTuneFdmWaveOcc, kTuneCount
};
inline int tuning_env(TuningKnob k)
{
    struct Table {
        int v[kTuneCount];
        Table()
        {
            static const char *const names[kTuneCount] = {"SMOQY_CHEB_WL0", "SMOQY_CHEB_SPLIT", "SMOQY_CHEB_OWN", "SMOQY_CHEB_GROUP", "SMOQY_FDM_STREAM", "SMOQY_FDM_OWN",
                                                          "SMOQY_FDM_OWN_MAX", "SMOQY_FDM_OWNSTREAM", "SMOQY_NT_FIELDS", "SMOQY_X_STREAM", "SMOQY_XCD_MAP", "SMOQY_TFFT_SLIM",
                                                          "SMOQY_TFFT_SB", "SMOQY_TFFT_EDGE", "SMOQY_CHEB_WAVE", "SMOQY_FDM_WAVE", "SMOQY_FDM_WAVE_R", "SMOQY_DMDX_FAST", "SMOQY_FDM_WAVE_OCC"};
            for (int q = 0; q < kTuneCount; ++q) {
                const char *e = getenv(names[q]);
                v[q] = e ? atoi(e) : -1;
            }
        }
    };
    static const Table t;
    return t.v[k];
}

}  // namespace smoqy
