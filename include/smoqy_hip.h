/*
 * smoqy_hip.h — C ABI of the MI355X-native (gfx950, HIP) CG / stochastic-trace hot path of
 * SmoQyElPhQMC.jl.
 *
 * The reference has no FFI: its "operator API" is Julia multiple dispatch on
 * FermionDetMatrix / KPMPreconditioner / PFFCalculator.  Each entry point below cites the
 * reference method (file:line under /root/reference) it replaces; INTEGRATION.md shows the
 * `ccall` shim a maintainer would add on the Julia side.
 *
 * Conventions at this boundary:
 *   - every function returns 0 on success, non-zero on error; smoqy_last_error() gives text.
 *     CG non-convergence is NOT an error (iters == maxiter is returned, as in the reference).
 *   - host arrays are column-major: state vectors are Ltau x N complex128 (interleaved re,im;
 *     tau contiguous), `count` vectors are stacked back to back (an Ltau x N x count array);
 *     fields are Ltau x N / Ltau x Nh float64.  Indices are Int64 and 1-based.
 *   - the library never keeps a host pointer after a call returns.
 *   - one in-flight call per handle; a handle owns one HIP stream.
 *   - matrix-element type T: Float64 (is_complex_T = 0) or ComplexF64 (is_complex_T = 1, complex hoppings).  With complex T
 *     the hopping-derived arrays cross the boundary as complex128 (interleaved re, im) exactly where the reference holds a
 *     Matrix{T}: cosh_dtt / sinh_dtt of smoqy_update_fields / smoqy_get_fields (Ltau x Nh), t of
 *     smoqy_update_from_path_integral[_all] (Nh x Ltau), and the Lanczos start vectors of smoqy_precond_update[_all] (N complex
 *     deviates per walker: randn! on a Vector{ComplexF64}).  V, expV and Λ stay real.  Sym complex handles run register-resident operator and Chebyshev kernels with the
 *     complex bond factor (round 4; 1.4x the real-T iteration at 16 walkers); Asym complex handles, τ-chunks above 2 and the Lanczos bounds run on the generic kernels.
 *     Round 3: the force terms and the device-side update! from the phonon fields take complex T as well — t0 of smoqy_set_bare_model is
 *     complex128 (Nh) and the SSH couplings carry their imaginary parts in smoqy_couplings.s_alpha*_im.
 *
 * A handle carries `nwalkers` independent field sets (one FermionDetMatrix + Λ + KPM
 * preconditioner each) times `nrhs` right-hand sides per walker; system s belongs to walker
 * s / nrhs.  nwalkers = nrhs = 1 is exactly one reference FermionDetMatrix.
 *
 * Device-resident batch vectors ("vec ids") hold one state vector per system in the device's
 * own slice-major layout; the *_v entry points work on them without touching the host.
 */
#ifndef SMOQY_HIP_H
#define SMOQY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smoqy_ctx smoqy_ctx;

/* op codes for smoqy_lambda_apply*: mul_Λ!, ldiv_Λ!, mul_Λᵀ!, ldiv_Λᵀ!
 * (src/holstein_shift_matrix.jl:47, 74, 102, 129) */
enum { SMOQY_LAMBDA_MUL = 0, SMOQY_LAMBDA_LDIV = 1, SMOQY_LAMBDA_MULT = 2, SMOQY_LAMBDA_LDIVT = 3 };

/* matvec selectors: mul_M!, mul_Mt!, mul_MtM! (= mul!), mul_MMt!
 * (src/FermionDetMatrix.jl:385/430, 484/528, 329, 357) */
enum { SMOQY_OP_M = 0, SMOQY_OP_MT = 1, SMOQY_OP_MTM = 2, SMOQY_OP_MMT = 3 };

/* ---- lifetime ------------------------------------------------------------------------- */

/* Sym/AsymFermionDetMatrix constructor (src/FermionDetMatrix.jl:66-111, 159-204) minus the
 * checkerboard decomposition, which is an INPUT: neighbor_table is 2 x Nh (1-based, colour
 * sorted), color_ranges is 2 x ncolors (1-based inclusive first/last bond of each colour).
 * Bonds of one colour must touch disjoint sites (checked).  device_id < 0 = current device. */
/* Lattices of more than 2556 sites run on generic kernels that stage their time slices in global memory instead of LDS
 * (functional, not tuned).  is_complex_T != 0 selects T = ComplexF64: bond factor [[c, s], [conj(s), c]]
 * (src/checkerboard_matrix_multiply.jl:60-68), s = sign(conj t) sinh(Δτ'|t|) (src/FermionDetMatrix.jl:224-231). */
int smoqy_create(smoqy_ctx **out, int Ltau, int N, int Nh, int ncolors, const int64_t *neighbor_table,
                 const int64_t *color_ranges, int is_sym, int is_complex_T, int nwalkers, int nrhs, int device_id);
int smoqy_destroy(smoqy_ctx *ctx);
/* a second handle on the same lattice, propagator form, device and preconditioner configuration with `nrhs` right-hand sides per walker
 * (the GreensEstimator's follower handle: Nrv systems per walker, its fields copied over with smoqy_copy_fields) */
int smoqy_clone(smoqy_ctx **out, const smoqy_ctx *src, int nrhs);
/* text of the last error on this handle (or of the last failed smoqy_create when ctx == NULL) */
const char *smoqy_last_error(const smoqy_ctx *ctx);
/* adopt a caller-owned hipStream_t (pass NULL to return to the handle's own stream) */
int smoqy_set_stream(smoqy_ctx *ctx, void *hip_stream);
int smoqy_sync(smoqy_ctx *ctx);
/* optional: page-locked host memory for arrays that are handed to the library repeatedly */
int smoqy_host_alloc(smoqy_ctx *ctx, void **ptr, size_t bytes);
int smoqy_host_free(smoqy_ctx *ctx, void *ptr);
/* page-lock memory the caller already owns (a Julia array, a shared-memory segment) / undo it */
int smoqy_host_register(smoqy_ctx *ctx, void *ptr, size_t bytes);
int smoqy_host_unregister(smoqy_ctx *ctx, void *ptr);
/* size(fdm) (src/FermionDetMatrix.jl:243): dims = {Ltau, N, Nh, ncolors, nwalkers, nrhs} */
int smoqy_dims(const smoqy_ctx *ctx, int dims[6]);

/* traits = {is_sym, is_complex_T, register-resident KPM kernels in use, wave-local colour-0 exchange (0 off, 1 ds_bpermute, 2 / 3 DPP),
 * one-wavefront-per-chain Chebyshev program (0 none, 1 ring, 2 plaquette), its lanes, register-resident operator kernels in use,
 * every colour a perfect matching}: what the handle's geometry selected at smoqy_create (tests and bench records read it) */
int smoqy_traits(const smoqy_ctx *ctx, int traits[8]);
/* JSON object naming the kernel families the handle's last full-batch fused MtM / Chebyshev launches ran and its tau-FFT form */
int smoqy_describe(const smoqy_ctx *ctx, char *buf, size_t n);

/* tau-chunk of the slice kernels (time slices per workgroup); Tc <= 0 restores the heuristic */
int smoqy_set_tau_chunk(smoqy_ctx *ctx, int Tc);
int smoqy_get_tau_chunk(const smoqy_ctx *ctx, int *Tc);

/* ---- fields --------------------------------------------------------------------------- */

/* overwrite expnΔτV, coshΔτt, sinhΔτt of one walker (fields of src/FermionDetMatrix.jl:46-48) */
int smoqy_update_fields(smoqy_ctx *ctx, int walker, const double *expV, const double *cosh_dtt, const double *sinh_dtt);
/* update!(fdm, fpi) on the device (src/FermionDetMatrix.jl:208-236): V is N x Ltau, t is
 * Nh x Ltau (FermionPathIntegral layout), perm the 1-based checkerboard permutation */
int smoqy_update_from_path_integral(smoqy_ctx *ctx, int walker, const double *V, const double *t, const int64_t *perm, double dtau);
/* the same for every walker of the handle in one pass: V_all is N x Ltau x nwalkers, t_all is
 * Nh x Ltau x nwalkers.  V_all == NULL or t_all == NULL leaves that part of the fields unchanged
 * (a walker whose hoppings do not depend on the phonon field need not resend them). */
int smoqy_update_from_path_integral_all(smoqy_ctx *ctx, const double *V_all, const double *t_all, const int64_t *perm, double dtau);
/* read the fields back (field access .expnΔτV etc., used by KPMPreconditioner.jl:208-209) */
int smoqy_get_fields(smoqy_ctx *ctx, int walker, double *expV, double *cosh_dtt, double *sinh_dtt);

/* ---- device-resident batch vectors ---------------------------------------------------- */

int smoqy_vec_alloc(smoqy_ctx *ctx, int *id);
int smoqy_vec_free(smoqy_ctx *ctx, int id);
/* host (Ltau x N x count, reference layout) <-> systems [sys0, sys0+count) of vector `id` */
int smoqy_vec_upload(smoqy_ctx *ctx, int id, const void *host, int sys0, int count);
int smoqy_vec_download(smoqy_ctx *ctx, int id, void *host, int sys0, int count);
int smoqy_vec_copy(smoqy_ctx *ctx, int dst, int src);
/* out[s] = dot(a_s, b_s) (conjugate-linear in a, like LinearAlgebra.dot); out is nsys complex128 */
int smoqy_vec_dot(smoqy_ctx *ctx, int a, int b, void *out);

/* ---- FermionDetMatrix applies --------------------------------------------------------- */

/* out = op(in) for every system; out == in allowed (lmul_M!/lmul_Mt!, :372, :470) */
int smoqy_matvec_v(smoqy_ctx *ctx, int op, int out, int in);
/* testing aid: route the applies through the generic kernels (any colouring / Asym path) */
int smoqy_matvec_force_generic(smoqy_ctx *ctx, int on);
/* kernel choice for mul_MtM! (src/FermionDetMatrix.jl:329-340) on Sym handles with real hoppings: run_len = 0 keeps one workgroup per τ-chunk;
 * run_len >= 2 makes every workgroup walk a run of that many time slices with its loads two slices ahead of the stage chain
 * (fdm_stream_kernel); run_len = -1 (the default) lets the library choose: streaming from 16 systems per launch, run length by launch
 * size.  Same stage order per site (outputs equal to rounding, FMA contraction aside); the p·Ap partial is formed as |M p|² instead of
 * p·(MᵀM p) (equal up to rounding, real and non-negative by construction). */
int smoqy_matvec_stream(smoqy_ctx *ctx, int run_len);
/* the same choice for the one-wavefront-per-run form (fdm_wave_kernel: the whole time slice in the registers of one wavefront, no LDS
 * image, no barrier) that handles take when their decomposition has a lane program — rings of up to 256 sites (2 colours), plaquette
 * lattices of up to 256 sites (4 colours), honeycomb lattices of up to 512 sites with hoppings that are τ-independent and equal on the
 * bonds of a colour (3 colours): run_len = -1 automatic (the default), 0 never, R >= 1 wavefronts walk runs of R slices (rounded down to
 * a multiple of the τ-chunk).  smoqy_describe names the kernel the last full-batch launch ran. */
int smoqy_matvec_wave(smoqy_ctx *ctx, int run_len);
/* host form: `count` vectors starting at system sys0 (fields of walker sys/nrhs); out == in allowed */
int smoqy_matvec(smoqy_ctx *ctx, int op, void *out, const void *in, int sys0, int count);

/* checkerboard_lmul! (inverse = 0) / checkerboard_ldiv! (inverse = 1) with the `transposed` flag and a
 * colour interval [color_first, color_first + ncolors) (0-based), in place
 * (src/checkerboard_matrix_multiply.jl:26-72, 98-145; `interval = checkerboard_colors[color]` at
 * src/fermion_det_matrix_dervative.jl:54-62) */
int smoqy_checkerboard_v(smoqy_ctx *ctx, int id, int inverse, int transposed, int color_first, int ncolors);
int smoqy_checkerboard(smoqy_ctx *ctx, void *inout, int inverse, int transposed, int color_first, int ncolors, int sys0, int count);

/* ---- Holstein shift matrix Λ ---------------------------------------------------------- */

/* set Λ (Ltau x N) of one walker from the host (what update_Λ! produced, :2-44) */
int smoqy_lambda_set(smoqy_ctx *ctx, int walker, const double *Lambda);
/* update_Λ! on the device (src/holstein_shift_matrix.jl:2-44): x is Nph x Ltau; ncoup
 * couplings with 1-based phonon / site ids; ph_sym[c] != 0 marks particle-hole symmetric form */
int smoqy_lambda_update(smoqy_ctx *ctx, int walker, const double *x, int Nph, double dtau, int ncoup,
                        const int64_t *coupling_to_phonon, const int64_t *coupling_to_site,
                        const double *alpha, const double *alpha3, const int32_t *ph_sym);
/* every walker in one pass (same coupling tables for all); x_all is Nph x Ltau x nwalkers */
int smoqy_lambda_update_all(smoqy_ctx *ctx, const double *x_all, int Nph, double dtau, int ncoup, const int64_t *coupling_to_phonon,
                            const int64_t *coupling_to_site, const double *alpha, const double *alpha3, const int32_t *ph_sym);
int smoqy_lambda_get(smoqy_ctx *ctx, int walker, double *Lambda);
/* mul_Λ!/ldiv_Λ!/mul_Λᵀ!/ldiv_Λᵀ! with the walker's stored Λ; out == in allowed */
int smoqy_lambda_apply_v(smoqy_ctx *ctx, int op, int out, int in);
/* host form with Λ passed by the caller; uses walker sys0/nrhs's slot as scratch */
int smoqy_lambda_apply(smoqy_ctx *ctx, int op, void *out, const void *in, const double *Lambda, int sys0, int count);

/* ---- FourierTransformer (src/FourierTransformer.jl:39-64) ----------------------------- */

int smoqy_fft_forward_v(smoqy_ctx *ctx, int id); /* lmul!(U, v): tau -> omega, unitary, antiperiodic */
int smoqy_fft_inverse_v(smoqy_ctx *ctx, int id); /* ldiv!(U, v) */
/* testing aid: 1 = use the rocFFT plans instead of the library's own fused tau-FFT kernels (which
 * are the default whenever Ltau factors into 2, 3, 5, 7) */
int smoqy_fft_use_rocfft(smoqy_ctx *ctx, int on);
int smoqy_fft_forward(smoqy_ctx *ctx, void *inout, int sys0, int count);
int smoqy_fft_inverse(smoqy_ctx *ctx, void *inout, int sys0, int count);

/* ---- KPMPreconditioner ---------------------------------------------------------------- */

/* KPMPreconditioner keyword arguments (src/KPMPreconditioner.jl:198-206); call before the first
 * update (defaults rbuf = 0.10, n = 20, a1 = 1.0, a2 = 1.0; a1 is doubled for Sym, :263) */
int smoqy_precond_config(smoqy_ctx *ctx, double rbuf, int n_lanczos, double a1, double a2);
/* update_preconditioner! (src/KPMPreconditioner.jl:554-597) for one walker.  randvec holds the
 * N normal deviates the caller's rng produces at :634 (the rng stays on the host). */
int smoqy_precond_update(smoqy_ctx *ctx, int walker, const double *randvec);
/* the same for every walker of the handle in one pass; randvecs is N x nwalkers */
int smoqy_precond_update_all(smoqy_ctx *ctx, const double *randvecs);
/* testing aid: route the KPM kernels through the generic (any colouring) fallback instead of the
 * register-resident fast path */
int smoqy_precond_force_generic(smoqy_ctx *ctx, int on);
/* state readback: active flag, bounds[2], order[] (cld(Ltau,2) entries for Sym, Ltau for Asym;
 * returns the count in *norder), Lanczos alpha[n] / beta[n-1].  Any pointer may be NULL. */
int smoqy_precond_get(smoqy_ctx *ctx, int walker, int *active, double *bounds, int *order, int *norder, double *lanczos_alpha, double *lanczos_beta);
/* coefficients of one frequency slot as complex128[order[slot]] */
int smoqy_precond_get_coefs(smoqy_ctx *ctx, int walker, int slot, void *coefs);
/* host-supplied state instead of smoqy_precond_update: bounds, order and the
 * concatenated complex128 coefficients of all slots; B̄ is still taken from the current fields */
int smoqy_precond_set(smoqy_ctx *ctx, int walker, int active, const double *bounds, const int *order, const void *coefs);
/* ldiv!(u', P, u) complex method (Sym :355-414, Asym :488-550); identity while inactive */
int smoqy_precond_apply_v(smoqy_ctx *ctx, int out, int in);
int smoqy_precond_apply(smoqy_ctx *ctx, void *out, const void *in, int sys0, int count);
/* ldiv!(u', P, u) real-vector methods (Sym :288-352, Asym :417-485): out / in are Ltau x N x count float64.  Half the frequencies are
 * evaluated, the rest is their complex conjugate (:334 / :468), the real part of the back-transform is returned (:344 / :475). */
int smoqy_precond_apply_real(smoqy_ctx *ctx, double *out, const double *in, int sys0, int count);

/* ---- conjugate gradient (ldiv!(v', fdm, v), src/FermionDetMatrix.jl:248-288) ----------- */

/* cg_solve! (src/IterativeSolvers/ConjugateGradient.jl:93-167 without, :169-249 with
 * preconditioner) of MᵀM x = b for every system.  x == b (same id) is the `x === b` case: zero
 * initial guess, b overwritten by the solution; otherwise x is the warm start.  iters / eps are
 * nsys entries.  use_precond != 0 applies each walker's KPM preconditioner (call
 * smoqy_precond_update first, as ldiv! does at FermionDetMatrix.jl:259). */
int smoqy_cg_solve_v(smoqy_ctx *ctx, int x, int b, double tol, int maxiter, int use_precond, int *iters, double *eps);
/* host form; x_is_b != 0 ignores the contents of x on entry */
int smoqy_cg_solve(smoqy_ctx *ctx, void *x, const void *b, int x_is_b, int sys0, int count, double tol, int maxiter, int use_precond, int *iters, double *eps);
/* process-wide gate for multi-threaded callers that drive several handles on one GPU: at most max_concurrent of them are inside a CG
 * solve at once (0 = no limit, the default); everything around the solves still overlaps.  Not tied to a handle. */
int smoqy_cg_gate(int max_concurrent);
/* two-part pipeline inside one handle: the systems of a batch are independent, so the iteration kernels of one half run on the handle's
 * stream and those of the other half on a second stream of the handle; the halves drift out of phase and overlap (what two handles on two
 * host threads do).  parts = 0: automatic (two parts from 8 systems up, the default), 1: off, 2..4: that many parts.  Results agree to rounding with identical
 * iteration counts (bit for bit when the parts select the same kernel family as the whole batch). */
int smoqy_cg_split(smoqy_ctx *ctx, int parts);
/* form of the library's own antiperiodic τ-FFT (FourierTransformer.jl:39-64 inside the preconditioner and the fused CG kernels):
 * in_place = 0 (default) keeps two images of a tile in LDS (three passes at Lτ = 128; fastest for one handle), 1 transforms in place
 * (half the LDS, two thirds of the registers, four passes: more workgroups per CU, +2.7 % sweeps/s with six handles on one GPU).
 * Same results to rounding (different pass order: a solve whose residual lands on the tolerance may take one more iteration); lengths
 * with a factor 7 keep the two-image form.  Time extents with a register-blocked kernel (one LDS pass: Lτ = 64, 128 always; Lτ = 80,
 * 100, 200 up to 64 systems per launch, and below 32 once in_place = 1 has said that several handles share the GPU) run that kernel
 * whatever is asked for here; smoqy_describe names the form in use. */
int smoqy_tfft_form(smoqy_ctx *ctx, int in_place);
/* host <-> device convergence polling period of the on-device CG loop (iterations per poll) */
int smoqy_cg_config(smoqy_ctx *ctx, int check_every);
/* replay one captured CG iteration as a hipGraph instead of launching its kernels one by one (off by
 * default: not faster on MI355X at the sizes measured) */
int smoqy_cg_use_graph(smoqy_ctx *ctx, int on);
/* a failed capture is not silent: the solve continues with eager launches, the switch is cleared, and the reason is what
 * smoqy_last_error returns after this call.  *enabled = the switch now, *captured = cached graphs that are still valid (the
 * cache is dropped whenever something baked into the captured kernel arguments changes: the coefficient table growing in
 * smoqy_precond_update/_set, smoqy_fft_use_rocfft, smoqy_set_tau_chunk, smoqy_set_stream, the *_force_generic switches). */
int smoqy_cg_graph_status(smoqy_ctx *ctx, int *enabled, int *captured);

/* ---- force terms of the pseudofermion action (handles with nrhs = 1) ------ */

/* Flattened electron-phonon couplings, i.e. what mul_νRe∂M∂x! / mul_νRe∂Λ∂x! read from
 * ElectronPhononParameters (src/fermion_det_matrix_dervative.jl:207-213, 257-262;
 * src/holstein_shift_matrix.jl:165-166).  Ids are 1-based.  s_bond[c] is the position n of coupling
 * c's hopping in the colour-sorted neighbour table (checkerboard_perm[n] == hopping,
 * hopping_to_couplings[hopping] ∋ c).  finite_mass[p] = isfinite(M[p]). */
typedef struct {
    int Nph;
    double dtau;
    const int32_t *finite_mass;
    int Nholstein;
    const double *h_alpha, *h_alpha2, *h_alpha3, *h_alpha4;
    const int64_t *h_coupling_to_phonon, *h_coupling_to_site;
    const int32_t *h_ph_sym;
    int Nssh;
    const double *s_alpha, *s_alpha2, *s_alpha3, *s_alpha4;
    const int64_t *s_coupling_to_phonon; /* 2 x Nssh */
    const int64_t *s_bond;
    /* T = ComplexF64 only (ssh_parameters.α::Vector{T}): imaginary parts of the SSH couplings, Nssh doubles each; NULL = real couplings.
     * Ignored by real handles. */
    const double *s_alpha_im, *s_alpha2_im, *s_alpha3_im, *s_alpha4_im;
} smoqy_couplings;

int smoqy_force_set_couplings(smoqy_ctx *ctx, const smoqy_couplings *cp);
/* phonon fields of every walker, Nph x Ltau x nwalkers (call after each move of x) */
int smoqy_force_set_phonons(smoqy_ctx *ctx, const double *x_all);
/* mul_νRe∂M∂x!(out, ν, u, v, fdm, elph) (src/fermion_det_matrix_dervative.jl:2-186): out is
 * Nph x Ltau x nwalkers on the host and is accumulated into, like the reference's ∂Sf∂x */
int smoqy_force_dMdx_v(smoqy_ctx *ctx, double nu, int u, int v, double *out);
/* mul_νRe∂Λ∂x!(out, ν, u′, u, Λ, elph) (src/holstein_shift_matrix.jl:156-201) with the stored Λ */
int smoqy_force_dLdx_v(smoqy_ctx *ctx, double nu, int up, int u, double *out);
/* the tail of calculate_derivative_fermionic_action! (src/PFFCalculator.jl:146-155) from Ψ:
 * ΛΨ, AΨ = MΛΨ, out += -2 Re<AΨ|∂M/∂x|ΛΨ>, MᵀAΨ, out += -2 Re<MᵀAΨ|∂Λ/∂x|Ψ> */
int smoqy_force_v(smoqy_ctx *ctx, int psi, double *out);
/* the same force, stored (not accumulated) into out — the first line of the HMC step is fill!(∂S∂x, 0)
 * (src/EFAPFFHMCUpdater.jl:160), so the transfer can land in the caller's array directly */
int smoqy_force_store_v(smoqy_ctx *ctx, int psi, double *out);

/* calculate_derivative_fermionic_action! (src/PFFCalculator.jl:119-157) for every walker of the handle in ONE call, with the
 * field update of the HMC step in front of it: [x_all != NULL: smoqy_update_from_phonons_all]; [randvec_all != NULL and
 * use_precond: update_preconditioner!, src/FermionDetMatrix.jl:259]; Ψ = Λ⁻ᵀΦ; Ψ = (MᵀM)⁻¹Ψ from a zero initial guess (:99);
 * Ψ = Λ⁻¹Ψ; Sf[w] = Re Φ·Ψ (:109); [dSdx != NULL: the force of smoqy_force_store_v].  Φ lives in vector phi, Ψ is left in
 * vector psi.  One host synchronisation at the end besides those of the preconditioner update and the CG's convergence polls. */
int smoqy_pff_step_v(smoqy_ctx *ctx, int phi, int psi, const double *x_all, const double *randvec_all, double tol, int maxiter, int use_precond,
                     double *Sf, int *iters, double *eps, double *dSdx);

/* ---- EFA leapfrog on the device ------------------------------------ */

/* SmoQyDQMC's ExactFourierAccelerator is NOT part of the reference tree: these entry points implement what its call sites in
 * src/EFAPFFHMCUpdater.jl fix — initialize_momentum! (:142), evolve_eom! (:150, :202), kinetic_energy (:244), the momentum kick
 * `p -= Δt ∂S∂x` (:196) — as exact harmonic evolution of every τ-Fourier mode of the phonon fields, with the accelerator's tables as
 * INPUTS.  Parity unpinned (no source to compare with); oracle: oracle/efa.py, pinned by energy conservation / reversibility / a dense
 * matrix exponential.  x, p and the force stay on the device across a whole trajectory.  Needs smoqy_force_set_couplings and
 * smoqy_set_bare_model, a handle with nrhs = 1 and Ltau = 2^a 3^b 5^c 7^d.
 *
 * q, m: Nph x Ltau (column-major, like x).  q[p, ω] = eigenvalue of the harmonic bosonic action of phonon p at the periodic τ-frequency ω
 * (S_b = 1/2 Σ q |x̃|², unitary transform), e.g. Δτ M [Ω² + 4/Δτ² sin²(πω/Lτ)]; m[p, ω] = dynamical mass of the fictitious momentum
 * (K = 1/2 Σ |p̃|²/m; m = q is "exact" acceleration: every mode turns with unit frequency).  Both must be symmetric under ω → Lτ-ω.
 * Phonons flagged infinite-mass in smoqy_couplings.finite_mass are left untouched. */
int smoqy_efa_config(smoqy_ctx *ctx, const double *q, const double *m);
/* x_all / p_all: Nph x Ltau x nwalkers; NULL leaves that array as it is.  Setting x also refreshes the fields (update!). */
int smoqy_efa_set_state(smoqy_ctx *ctx, const double *x_all, const double *p_all);
int smoqy_efa_get_state(smoqy_ctx *ctx, double *x_all, double *p_all);
/* initialize_momentum!(p, efa, rng): R_all holds Nph x Ltau x nwalkers unit normal deviates from the caller's rng; p = F⁻¹ √m F R;
 * K[w] = kinetic energy (src/EFAPFFHMCUpdater.jl:142) */
int smoqy_efa_initialize_momentum(smoqy_ctx *ctx, const double *R_all, double *K);
/* kinetic_energy(p, efa) (:244) and the harmonic bosonic action 1/2 Σ q |x̃|² of the current state, per walker (either may be NULL) */
int smoqy_efa_energies(smoqy_ctx *ctx, double *K, double *Sb);
/* [p -= kick_dt * ∂S/∂x with the force the last force evaluation left on the device (:196)]; evolve_eom!(x, p, dt, efa) (:150, :202);
 * [refresh_fields != 0: update!(fermion_path_integral), update!(fdm), update_Λ! from the new x (:203-205)] */
int smoqy_efa_evolve(smoqy_ctx *ctx, double dt, double kick_dt, int refresh_fields);
/* restore = 0: copyto!(x0, x) (:130);  restore = 1: the reject branch — copyto!(x, x0) and update! (:263-275) */
int smoqy_efa_checkpoint(smoqy_ctx *ctx, int restore);
/* the reject branch for some walkers of the batch only: restore[w] != 0 puts walker w back to its checkpoint (every replica takes its own
 * Metropolis decision, :252-275), the others keep the fields the trajectory left */
int smoqy_efa_restore_walkers(smoqy_ctx *ctx, const int *restore);
/* the trajectory of hmc_update! between the momentum refresh and the final action (src/EFAPFFHMCUpdater.jl:148-206) in ONE call:
 * evolve(Δt/2), update!; Nt times { calculate_derivative_fermionic_action! at tol_force; p -= Δt ∂S_f/∂x; evolve(Δt, last step Δt/2);
 * update! }.  Φ in vector phi, Ψ left in psi.  randvecs: N x nwalkers x Nt Lanczos start vectors (2N each for a complex handle; the rng stays on the host; sent in one transfer); Sf, iters,
 * eps: nwalkers x Nt (any may be NULL).  Only the fermionic force is applied: anharmonic / dispersive phonon terms (:190-193) are the
 * caller's to add through smoqy_efa_evolve step by step.  ASSUMES recenter! = identity — the default of hmc_update! (:112) and what every
 * shipped script passes: the reference calls recenter!(x) after each evolve_eom! (:151, :203), which this call does not; a driver with
 * another recenter! keeps the step-by-step form (smoqy_efa_evolve, then its recenter! on the host copy, smoqy_efa_set_state).  A non-zero return (e.g. a non-finite residual) leaves x and p wherever the
 * trajectory stopped: the caller rejects the update with smoqy_efa_checkpoint(ctx, 1), as the reference's catch block does (:176-187). */
int smoqy_hmc_trajectory_v(smoqy_ctx *ctx, int phi, int psi, int Nt, double dt, double tol_force, int maxiter, int use_precond, const double *randvecs, double *Sf, int *iters, double *eps);
/* How smoqy_hmc_trajectory_v waits.  on = 0: for every force solve (a poll of the CG states per step).  on = 1 (the default): not at all
 * until the trajectory's end — each solve is launched with the iterations its step needed in the previous trajectory of the same length
 * and tolerance plus a margin, the per-step states stay on the device, and at the end EVERY solve is checked (converged, finite residual);
 * if one is not, x, p and the fields are put back and the trajectory is repeated with polls.  Same kernels, same iterations: the results
 * of the two forms are identical.  The library takes the asynchronous form only where it pays: every step of the previous trajectory
 * within 64 iterations (longer solves gain nothing from the missing polls and their counts move by tens), and after a repeated trajectory
 * the next 1, 2, 4 … 64 ones poll.  on < 0 leaves the setting alone; *runs / *misses (may be NULL) count the asynchronous trajectories
 * and the ones that had to be repeated. */
int smoqy_hmc_async(smoqy_ctx *ctx, int on, long *runs, long *misses);

/* ---- device-side update! from the phonon fields ------------------- */

/* bare on-site energies V⁰ (N) and hoppings t⁰ (Nh, FermionPathIntegral order) — what
 * SmoQyDQMC.update!(fermion_path_integral, elph, x, -1) leaves behind (src/EFAPFFHMCUpdater.jl:148, 200);
 * perm is the 1-based checkerboard permutation.  Needs smoqy_force_set_couplings.  T = ComplexF64: t0 is complex128 (interleaved re, im). */
int smoqy_set_bare_model(smoqy_ctx *ctx, const double *V0, const double *t0, const int64_t *perm);
/* SmoQyDQMC.update!(fermion_path_integral, elph, x, +1); update!(fdm, fpi); update_Λ! for every walker
 * from ONE upload of x (Nph x Ltau x nwalkers): V = V⁰ + Σ(αx+α₂x²+α₃x³+α₄x⁴), t = t⁰ - Σ(αΔx+…),
 * then src/FermionDetMatrix.jl:208-236 and src/holstein_shift_matrix.jl:2-44 on the device
 * (call sites src/EFAPFFHMCUpdater.jl:148-152, 200-205).  Also refreshes the force kernels' x. */
int smoqy_update_from_phonons_all(smoqy_ctx *ctx, const double *x_all);

/* ---- GreensEstimator ------------------------------------------------ */

/* copy one walker's fields (expnΔτV, cosh, sinh, Λ) from another handle of the same lattice on the same device:
 * a measurement handle created with nrhs = Nrv follows the sampling handle without a host round trip
 * (update_greens_estimator!(ge, fermion_det_matrix, …), src/Measurements/GreensEstimator.jl:125-175) */
int smoqy_copy_fields(smoqy_ctx *dst, int dst_walker, smoqy_ctx *src, int src_walker);
/* unit cell and lattice of the model geometry (GreensEstimator constructor :76-99): n orbitals per cell,
 * D <= 2 directions of extent L[d]; site = orbital + n * cell, cell = c1 + L1 * c2 (column-major) */
int smoqy_ge_config(smoqy_ctx *ctx, int n_orbitals, int D, const int64_t *L);
/* measure_GΔ0!(…, greens_estimator, (a, b)) (:179-233): G(Δ,0) averaged over translations and over the
 * handle's nrhs random vectors, from GR = M⁻¹R (vector gr) and R (vector r, conjugated on the fly as Rt).
 * out is complex (Lτ+1) x L... x nwalkers, the reference's CΔ0 array per walker, before
 * add_contraction_to_correlation! permutes τ to the last axis (:712-726).  a, b are 1-based orbitals. */
int smoqy_ge_measure_GD0(smoqy_ctx *ctx, int gr, int r, int a, int b, void *out);

/* one factor of a four-point estimator: which solved/random vector, which orbital, the static displacement of
 * ShiftedArrays.circshift(·, (0, -r..., 0)) (src/Measurements/GreensEstimator.jl:265-268), and whether it takes the
 * first (n) or the second (m) random vector of a pair n < m */
typedef struct {
    int source;        /* 0: GR = M⁻¹R, 1: Rt = conj(R) */
    int orbital;       /* 1-based */
    int64_t shift[2];  /* r, unit cells */
    int second;        /* 0: vector n of the pair, 1: vector m */
} smoqy_ge_slot;
/* the pair sum shared by measure_GΔ0_GΔ0!, measure_GΔΔ_G00! and measure_G0Δ_GΔ0! (:285-306, 439-460, 518-539):
 *   CΔ0 = 1/Npairs Σ_{n<m} _measure_CΔ0!(slot0 ⊙ slot1 [⊙ tΔ], slot2 ⊙ slot3 [⊙ t0])        (:610-652, 677-708)
 * over all pairs of the handle's nrhs random vectors, per walker.  tD / t0 are complex (Lτ x L...) weight arrays or
 * NULL, conjugated when the flag is set (bconj, :729).  out is complex (Lτ+1) x L... x nwalkers; the scalar boundary
 * terms at τ = 0 / τ = β (:312-382, 545-599) are left to the caller, who holds GR and Rt. */
int smoqy_ge_measure_pairs(smoqy_ctx *ctx, int gr, int r, const smoqy_ge_slot *slots, const void *tD, int conj_tD, const void *t0, int conj_t0, void *out);

/* the scalar those boundary terms subtract (e.g. :325-334): per walker
 *   1/(Nrv·Lτ·Ncells) Σ_rv Σ_{τ,c} [bconj(circshift(tΔ, tshift)) · bconj(t0)] · circshift(GR_orbital_gr, shift) · Rt_orbital_r
 * from the device-resident vectors (circshift(a, s)[c] = a[c - s]; tD = t0 = NULL drops the weights).  out: nwalkers complex. */
int smoqy_ge_boundary_dot(smoqy_ctx *ctx, int gr, int r, int orbital_gr, int orbital_r, const int64_t *shift, const void *tD, int conj_tD, const int64_t *tshift, const void *t0, int conj_t0, void *out);

/* ---- walker teams: the reference's one-walker-per-rank control flow on one batched handle ------------------ */

/* The reference runs each walker as its own MPI rank with single-walker objects (tutorials/holstein_honeycomb_mpi.jl:60-72).  On one GPU
 * that model does not scale (ranks time-slice the device); a handle with nwalkers = K does, but it needs the K control flows in lock step.
 * A team is that rendezvous: K host threads (one per replica) each run the UNCHANGED per-walker update sequence and call the entry points
 * below with their walker index w; a call blocks until all K members have made the same call, the last arrival runs the batched library
 * call for everybody, every member returns with its own results.  All members of a round must pass the same tol / maxiter / use_precond.
 * A non-zero return (e.g. a non-finite residual in any member's solve, or a member that never arrives: rendezvous time-out, code 9) is
 * delivered to every member of the round; each rejects its update, as the reference's catch block does (src/EFAPFFHMCUpdater.jl:168-187).
 * The handle needs smoqy_force_set_couplings and smoqy_set_bare_model (the field update from x runs on the device); Nph is the number of
 * phonon modes of those couplings.  While a team is in use, nothing else may call into its handle. */
typedef struct smoqy_team smoqy_team;
int smoqy_team_create(smoqy_team **out, smoqy_ctx *ctx, int Nph);
int smoqy_team_destroy(smoqy_team *team);
const char *smoqy_team_last_error(const smoqy_team *team);
int smoqy_team_size(const smoqy_team *team, int *K);
/* seconds a member waits for the others before it gives up with code 9 (default 600) */
int smoqy_team_set_timeout(smoqy_team *team, double seconds);
/* the device-resident Φ / Ψ of the team's PFFCalculator (src/PFFCalculator.jl:9-16), e.g. for smoqy_vec_download between rounds */
int smoqy_team_vectors(const smoqy_team *team, int *phi, int *psi);
/* sample_pseudofermion_fields! (src/PFFCalculator.jl:56-76) for member w: R is the member's Ltau x N complex normal deviates (randn! on
 * its own rng, :67); Φ = Λᵀ Mᵀ R stays on the device; *RdotR = |R|² */
int smoqy_team_sample_phi(smoqy_team *team, int w, const void *R, double *RdotR);
/* calculate_fermionic_action! (dSdx == NULL) / calculate_derivative_fermionic_action! (src/PFFCalculator.jl:79-157) for member w, with the
 * field update of the member's move in front: x = the member's Nph x Ltau phonon fields after its move (SmoQyDQMC.update! + update!(fdm) +
 * update_Λ! on the device, src/EFAPFFHMCUpdater.jl:200-205) or NULL if they did not change; randvec = N normal deviates for the Lanczos
 * start vector (KPMPreconditioner.jl:634; needed when use_precond != 0).  Out: S_f, (iters, eps) of the solve, ∂S_f/∂x (Nph x Ltau, stored). */
int smoqy_team_pff_step(smoqy_team *team, int w, const double *x, const double *randvec, double tol, int maxiter, int use_precond, double *Sf, int *iters, double *eps, double *dSdx);

/* hmc_update! (src/EFAPFFHMCUpdater.jl:102-276) of member w with the whole trajectory on the device (the team's handle needs
 * smoqy_efa_config): fields from x (NULL = unchanged), Φ = Λᵀ Mᵀ R (:133), copyto!(x0, x) (:130), momenta from the member's Nph x Ltau unit
 * normal deviates P (:142), the leapfrog of Nt steps (smoqy_hmc_trajectory_v, :148-206), the final action at `tol` (:217).  randvecs: N x (Nt+1)
 * Lanczos start vectors (one per force evaluation and one for the final action; Nt <= 64).  Out: H0 = {S_f, S_b, K} before the trajectory,
 * H1 = the same after it, x_new = the proposed fields (Nph x Ltau; may be NULL), *iters = CG iterations of all Nt + 1 solves.  The member
 * then takes its Metropolis decision (:247-260) and MUST report it with smoqy_team_hmc_finish before any other team call: accept != 0
 * keeps the proposed fields on the device, 0 restores this walker's checkpoint (:263-275).  All members of a round pass the same Nt, dt,
 * tolerances and maxiter.  A failing solve restores every member's fields and returns non-zero to everybody (the catch block, :176-187);
 * no finish call follows then. */
int smoqy_team_hmc_update(smoqy_team *team, int w, const double *x, const void *R, const double *P, const double *randvecs, int Nt, double dt, double tol_force, double tol, int maxiter,
                          double *H0, double *H1, double *x_new, int *iters);
int smoqy_team_hmc_finish(smoqy_team *team, int w, int accept);

/* GreensEstimator for team members (src/Measurements/GreensEstimator.jl): smoqy_team_ge_config — once, before smoqy_team_serve — clones the
 * team's handle with Nrv right-hand sides per walker (smoqy_clone) and configures the contractions (smoqy_ge_config: n_orbitals, D, L).
 * smoqy_team_ge_update = update_greens_estimator! (:125-175) of member w: R = the member's Ltau x N x Nrv unit-modulus random vectors
 * (randn!, R ./= abs.(R), :141-142), randvec = the Lanczos start vector of update_preconditioner! (:150); the follower handle takes every
 * walker's current fields and solves all K·Nrv systems in ONE batched CG; *iters = iterations summed over the member's Nrv solves, *eps = its
 * largest residual.  smoqy_team_ge_measure_GD0 = measure_GΔ0! (:179-233) for orbitals (a, b), 1-based, the same for all members of a round:
 * out = the member's (Ltau+1) x L... complex array. */
int smoqy_team_ge_config(smoqy_team *team, int Nrv, int n_orbitals, int D, const int64_t *L);
int smoqy_team_ge_update(smoqy_team *team, int w, const void *R, const double *randvec, double tol, int maxiter, int *iters, double *eps);
int smoqy_team_ge_measure_GD0(smoqy_team *team, int w, int a, int b, void *out);

/* Teams across processes — the reference's walkers are MPI ranks (processes): the rank that owns the GPU handle publishes its team in a
 * POSIX shared-memory segment `name` ("/something"); every rank of the node, the serving one included, joins with smoqy_member_attach and
 * makes the same two calls as a team member.  A member needs no GPU and no handle: it copies its arrays into the segment (page-locked in
 * the serving process) and sleeps until a server thread in the serving process has run the round for all K members.  x0 (optional): the
 * K members' initial phonon fields, Nph x Ltau each, which smoqy_member_fields hands to the members.  While a team is published its
 * in-process entry points (smoqy_team_sample_phi / _pff_step) are closed.  smoqy_team_unserve (also run by smoqy_team_destroy) wakes every
 * waiting member with code 10 and removes the segment. */
int smoqy_team_serve(smoqy_team *team, const char *name, const double *x0);
int smoqy_team_unserve(smoqy_team *team);
typedef struct smoqy_member smoqy_member;
/* waits up to wait_seconds for the segment to appear (the ranks of a job start together); w = this rank's walker index in the team */
int smoqy_member_attach(smoqy_member **out, const char *name, int w, double wait_seconds);
int smoqy_member_detach(smoqy_member *member);
const char *smoqy_member_last_error(const smoqy_member *member);
/* dims[0..3] = Ltau, N, K, Nph */
int smoqy_member_dims(const smoqy_member *member, int *dims);
/* the phonon fields the serving handle last received for this walker (x0 of smoqy_team_serve before the first step) */
int smoqy_member_fields(const smoqy_member *member, double *x);
/* smoqy_team_sample_phi / smoqy_team_pff_step for this member */
int smoqy_member_sample_phi(smoqy_member *member, const void *R, double *RdotR);
int smoqy_member_pff_step(smoqy_member *member, const double *x, const double *randvec, double tol, int maxiter, int use_precond, double *Sf, int *iters, double *eps, double *dSdx);
/* smoqy_team_hmc_update / smoqy_team_hmc_finish for this member */
int smoqy_member_hmc_update(smoqy_member *member, const double *x, const void *R, const double *P, const double *randvecs, int Nt, double dt, double tol_force, double tol, int maxiter,
                            double *H0, double *H1, double *x_new, int *iters);
int smoqy_member_hmc_finish(smoqy_member *member, int accept);
/* smoqy_team_ge_update / smoqy_team_ge_measure_GD0 for this member; _ge_dims: Nrv and the bytes of one member's G(Δ,0) array (0 / 0 if the
 * serving rank did not configure a GreensEstimator) */
int smoqy_member_ge_dims(const smoqy_member *member, int *Nrv, size_t *g_bytes);
int smoqy_member_ge_update(smoqy_member *member, const void *R, const double *randvec, double tol, int maxiter, int *iters, double *eps);
int smoqy_member_ge_measure_GD0(smoqy_member *member, int a, int b, void *out);

/* ---- measurement aids (bench.py) -------------------------------------------------------- */

/* K native member threads (K = the team's size) each run `warmup_sweeps` untimed and `nsweeps` timed sweeps of the reference tutorial's
 * per-walker update sequence (tutorials/holstein_honeycomb.jl:611-684: two local-move-like updates, an HMC trajectory of Nt force
 * evaluations, the closing action) through smoqy_team_sample_phi / smoqy_team_pff_step, drawing their own normal deviates: the number a
 * caller WITHOUT an interpreter lock gets from a team.  x0: the K members' phonon fields (Nph x Ltau each, member-major); the first
 * `nfree` modes of every slice are moved by the synthetic drift of bench.py's sweep and restored.  device_hmc != 0: the HMC part is
 * smoqy_team_hmc_update (trajectory on the device, Δt = π/(2 Nt), always rejected) instead of Nt host-driven force steps.  Out: wall
 * seconds of the timed part, solves and CG iterations summed over the members. */
int smoqy_team_bench_sweeps(smoqy_team *team, const double *x0, int nfree, double drift, int Nt, double tol, double tol_force, int maxiter, int device_hmc, int warmup_sweeps,
                            int nsweeps, unsigned long seed, double *seconds, long *solves, long *iters);

/* The normal deviates the native member threads draw (xoshiro256++ and a 128-layer ziggurat), exposed so that their distribution can be
 * tested on the CPU: n numbers of standard deviation `scale` from the stream of `seed`.  Host code only; needs no GPU and no handle. */
int smoqy_bench_randn(double *out, long n, unsigned long seed, double scale);

/* HIP events on the handle's stream */
int smoqy_timer_start(smoqy_ctx *ctx);
int smoqy_timer_stop(smoqy_ctx *ctx, double *ms);
/* `reps` back-to-back launches of one matvec kernel between two HIP events; *ms = total */
/* in-situ duration of the fused MᵀM launches inside the CG loop: every sample_every-th full-batch launch is bracketed
 * by an event pair on the handle's stream (at most max_samples); _read synchronises, returns the mean and stops sampling */
int smoqy_matvec_timing(smoqy_ctx *ctx, int sample_every, int max_samples);
int smoqy_matvec_timing_read(smoqy_ctx *ctx, double *avg_us, int *samples);
/* the same sampled launches by the device's constant 100 MHz clock: mean of (last workgroup's end - first workgroup's start), the
 * interval rocprofv3 --kernel-trace reports for a dispatch (an event pair on a busy stream also holds the gap to the previous
 * launch).  Call before smoqy_matvec_timing_read, which ends the sampling. */
int smoqy_matvec_timing_read_device(smoqy_ctx *ctx, double *avg_us, int *samples);
/* per-kernel durations of the fused CG iteration inside real solves: the next `iterations` full-batch iterations on the handle's stream get
 * an event in front of each of their four launches and one behind the last; _read returns us[0..3] = mean event-to-event time of the fused
 * MᵀM, the forward τ-FFT (with the r update), the Chebyshev apply and the inverse τ-FFT (with the x / p updates), and ends the sampling */
int smoqy_cg_iteration_timing(smoqy_ctx *ctx, int iterations);
int smoqy_cg_iteration_timing_read(smoqy_ctx *ctx, double *us, int *iterations);
/* device stream-copy ceiling: `reps` device-to-device copies of `bytes` bytes by a plain 16-byte-per-lane copy
 * kernel between two HIP events; each copy moves 2 * bytes.  Buffers are allocated and freed inside the call. */
int smoqy_bench_copy(smoqy_ctx *ctx, size_t bytes, int reps, double *ms);
int smoqy_bench_matvec(smoqy_ctx *ctx, int op, int out, int in, int reps, double *ms);
/* algorithmic bytes of one launch of `op` over all systems ((2S+F) per M / Mᵀ,
 * 2(2S+F) per MᵀM / MMᵀ, F counted once per walker) */
int smoqy_algorithmic_bytes(const smoqy_ctx *ctx, int op, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* SMOQY_HIP_H */
