"""Host-side mirror of the reference's operator API for the CG / stochastic-trace hot path.

Same names, argument meaning and error behaviour as the Julia functions (minus the ``!``), so
the parity tests read like the reference's own call sites.  Every function that computes goes
through the C ABI of ``libsmoqy_hip.so``; there is no CPU implementation behind these names.

Vectors are numpy ``complex128`` arrays in the reference layout (``Ltau x N`` column-major, i.e.
``order="F"``; a flat vector of length ``Ltau*N`` is accepted too, like ``reshaped`` at
src/SmoQyElPhQMC.jl:18-20).  Output arguments are written in place.

reference type / function                      here
---------------------------------------------  ----------------------------------------------
SymFermionDetMatrix / AsymFermionDetMatrix     same (src/FermionDetMatrix.jl:44-204)
update!(fdm, fpi)                              update(fdm, fpi)                 (:208-236)
mul_M!, mul_Mt!, mul_MtM!, mul_MMt!, mul!      mul_M, mul_Mt, mul_MtM, mul_MMt, mul
lmul_M!, lmul_Mt!, lmul_MtM!, lmul_MMt!, lmul! lmul_M, lmul_Mt, lmul_MtM, lmul_MMt, lmul
ldiv!(v', fdm, v; preconditioner, ...)         ldiv(vp, fdm, v, preconditioner=I, ...) (:248-288)
ConjugateGradientSolver, cg_solve!             same, cg_solve (IterativeSolvers/ConjugateGradient.jl)
KPMPreconditioner, update_preconditioner!      same, update_preconditioner (KPMPreconditioner.jl)
ldiv!(u', P, u)                                ldiv(up, P, u)
FourierTransformer, mul!/lmul!/ldiv!           same, mul / lmul / ldiv (FourierTransformer.jl)
update_Λ!, mul_Λ!, ldiv_Λ!, mul_Λᵀ!, ldiv_Λᵀ!   update_Λ, mul_Λ, ldiv_Λ, mul_Λᵀ, ldiv_Λᵀ (+ ASCII aliases)
PFFCalculator, sample_pseudofermion_fields!,   same (src/PFFCalculator.jl)
calculate_fermionic_action!
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib as L
from .lattice import ElectronPhononParameters, FermionPathIntegral, checkerboard_decomposition


class _Identity:
    """``LinearAlgebra.I`` — the default ``preconditioner = I``."""

    def __repr__(self):
        return "I"


I = _Identity()


class ConjugateGradientSolver:
    """src/IterativeSolvers/ConjugateGradient.jl:16-60.  The work vectors r, p, z live on the
    device inside the handle; this object carries ``maxiter``, ``tol`` and ``N``."""

    def __init__(self, z, maxiter=None, tol=1e-2):
        n = int(np.size(z))
        self.N = n
        self.maxiter = n if maxiter is None else int(maxiter)  # the reference default is the (buggy) length(z), :50
        self.tol = float(tol)


class FermionDetMatrix:
    """Abstract base (src/FermionDetMatrix.jl:19).  One instance owns one ``smoqy_ctx`` with a
    single walker; ``nrhs`` > 1 lets batched callers (GreensEstimator-style multi-RHS solves)
    share the fields."""

    is_sym = True

    def __init__(self, fermion_path_integral: FermionPathIntegral, maxiter=None, tol=1e-6, nrhs=1, device=-1):
        fpi = fermion_path_integral
        self.Lt, self.N = int(fpi.Ltau), int(fpi.N)
        self.Nh = int(np.shape(fpi.t)[0])
        maxiter = self.N * self.Lt if maxiter is None else int(maxiter)
        # checkerboard decomposition (Checkerboard.jl stand-in), :96 / :189
        self.checkerboard_neighbor_table, self.checkerboard_perm, colors = checkerboard_decomposition(fpi.neighbor_table)
        self._colors = colors
        self.checkerboard_colors = [range(int(colors[0, c]), int(colors[1, c]) + 1) for c in range(colors.shape[1])]
        self.cgs = ConjugateGradientSolver(np.empty(self.Lt * self.N), maxiter=maxiter, tol=tol)
        # matrix-element type T of FermionDetMatrix{T, E} (:19): complex when the path integral carries complex hoppings
        self.T = np.complex128 if np.iscomplexobj(fpi.t) else np.float64
        self.handle = L.Handle(self.Lt, self.N, self.checkerboard_neighbor_table, colors, self.is_sym, 1, nrhs, device, is_complex=self.T is np.complex128)
        self.nrhs = nrhs
        update(self, fpi)

    # field access used by KPMPreconditioner (:208-209) and the force code
    def _fields(self):
        e = np.zeros((self.Lt, self.N), order="F")
        c = np.zeros((self.Lt, self.Nh), order="F", dtype=self.T)  # Matrix{T}, :47-48
        s = np.zeros((self.Lt, self.Nh), order="F", dtype=self.T)
        self.handle.call("smoqy_get_fields", 0, L.ptr(e), L.ptr(c), L.ptr(s))
        return e, c, s

    @property
    def expnΔτV(self):
        return self._fields()[0]

    @property
    def coshΔτt(self):
        return self._fields()[1]

    @property
    def sinhΔτt(self):
        return self._fields()[2]


class SymFermionDetMatrix(FermionDetMatrix):
    """B_l = [e^{-ΔτK_l/2}]ᵀ e^{-ΔτV_l} e^{-ΔτK_l/2} (src/FermionDetMatrix.jl:22-111)."""

    is_sym = True


class AsymFermionDetMatrix(FermionDetMatrix):
    """B_l = e^{-ΔτV_l} e^{-ΔτK_l} (src/FermionDetMatrix.jl:115-204)."""

    is_sym = False


def update(fermion_det_matrix: FermionDetMatrix, fermion_path_integral: FermionPathIntegral):
    """update!(fdm, fpi), src/FermionDetMatrix.jl:208-236, computed on the device."""
    fpi = fermion_path_integral
    V = np.asfortranarray(fpi.V, dtype=np.float64)
    if np.iscomplexobj(fpi.t) and fermion_det_matrix.T is not np.complex128:
        raise TypeError("complex hoppings need a FermionDetMatrix constructed with matrix-element type T = ComplexF64")
    t = np.asfortranarray(fpi.t, dtype=fermion_det_matrix.T)
    if V.shape != (fermion_det_matrix.N, fermion_det_matrix.Lt) or t.shape != (fermion_det_matrix.Nh, fermion_det_matrix.Lt):
        raise ValueError("FermionPathIntegral arrays do not match the FermionDetMatrix")
    fermion_det_matrix.handle.call("smoqy_update_from_path_integral", 0, L.ptr(V), L.ptr(t), L.ptr(fermion_det_matrix.checkerboard_perm), C.c_double(fpi.dtau))


def size(fermion_det_matrix: FermionDetMatrix, dim=None):
    n = fermion_det_matrix.Lt * fermion_det_matrix.N
    return n if dim is not None else (n, n)


def eltype(fermion_det_matrix: FermionDetMatrix):
    return fermion_det_matrix.T


# ---- matrix-vector products --------------------------------------------------------------------

def _matvec(op, vp, fdm, v):
    h = fdm.handle
    vin = L.as_state(v, h.Lt, h.N)
    vout = L.writable_state(vp, h.Lt, h.N)
    h.call("smoqy_matvec", op, L.ptr(vout), L.ptr(vin), 0, 1)


def mul_M(vp, fdm, v):
    """mul_M!: v′ = M v (Sym :385-427, Asym :430-466)."""
    _matvec(L.OP_M, vp, fdm, v)


def mul_Mt(vp, fdm, v):
    """mul_Mt!: v′ = Mᵀ v (Sym :484-525, Asym :528-563)."""
    _matvec(L.OP_MT, vp, fdm, v)


def mul_MtM(vp, fdm, v):
    """mul_MtM!: v′ = Mᵀ M v (:329-340)."""
    _matvec(L.OP_MTM, vp, fdm, v)


def mul_MMt(vp, fdm, v):
    """mul_MMt!: v′ = M Mᵀ v (:357-368)."""
    _matvec(L.OP_MMT, vp, fdm, v)


def lmul_M(fdm, v):
    mul_M(v, fdm, v)


def lmul_Mt(fdm, v):
    mul_Mt(v, fdm, v)


def lmul_MtM(fdm, v):
    mul_MtM(v, fdm, v)


def lmul_MMt(fdm, v):
    mul_MMt(v, fdm, v)


# ---- checkerboard multiplies -----------------------------------------------------------------------

def _interval_to_colours(fdm, interval):
    """A reference `interval` is a bond range; the calls on the hot path and in the force code pass
    either everything or one colour (`checkerboard_colors[color]`).  Returns (first colour, count)."""
    if interval is None:
        return 0, len(fdm.checkerboard_colors)
    lo, hi = interval.start, interval.stop - 1
    first = [k for k, r in enumerate(fdm.checkerboard_colors) if r.start == lo]
    last = [k for k, r in enumerate(fdm.checkerboard_colors) if r.stop - 1 == hi]
    if not first or not last or last[0] < first[0]:
        raise ValueError("interval must be a union of consecutive checkerboard colours")
    return first[0], last[0] - first[0] + 1


def checkerboard_lmul(v, fdm, transposed=False, interval=None):
    """checkerboard_lmul!(v, neighbor_table, coshΔτt, sinhΔτt; transposed, interval)
    (src/checkerboard_matrix_multiply.jl:26-72) with the matrix's own tables."""
    c0, nc = _interval_to_colours(fdm, interval)
    a = L.writable_state(v, fdm.Lt, fdm.N)
    fdm.handle.call("smoqy_checkerboard", L.ptr(a), 0, int(transposed), c0, nc, 0, 1)


def checkerboard_ldiv(v, fdm, transposed=False, interval=None):
    """checkerboard_ldiv! (src/checkerboard_matrix_multiply.jl:98-145)."""
    c0, nc = _interval_to_colours(fdm, interval)
    a = L.writable_state(v, fdm.Lt, fdm.N)
    fdm.handle.call("smoqy_checkerboard", L.ptr(a), 1, int(transposed), c0, nc, 0, 1)


def checkerboard_mul(vp, v, fdm, transposed=False, interval=None):
    """checkerboard_mul!: copy then lmul (src/checkerboard_matrix_multiply.jl:2-22)."""
    o = L.writable_state(vp, fdm.Lt, fdm.N)
    o[...] = np.asarray(v).reshape(o.shape, order="F")
    checkerboard_lmul(o, fdm, transposed, interval)


# ---- FourierTransformer ------------------------------------------------------------------------------

class FourierTransformer:
    """src/FourierTransformer.jl:2-21.  ``θ`` as in the reference; the plans are rocFFT plans
    owned by a handle (an existing FermionDetMatrix's, or a private bond-less one)."""

    def __init__(self, T=np.float64, Lτ=None, N=None, handle=None):
        if isinstance(T, np.ndarray):  # FourierTransformer(v::Matrix)
            Lτ, N = T.shape
        self.Lτ, self.N = int(Lτ), int(N)
        self.θ = np.exp(-1j * np.pi * np.arange(self.Lτ) / self.Lτ)
        if handle is None:
            handle = L.Handle(self.Lτ, self.N, np.zeros((2, 0), dtype=np.int64), np.zeros((2, 0), dtype=np.int64), True, 1, 1, -1)
        self.handle = handle


# ---- KPM preconditioner ----------------------------------------------------------------------------

class KPMPreconditioner:
    """KPMPreconditioner(fdm; rng, rbuf, n, a1, a2), src/KPMPreconditioner.jl:198-284.  The state
    (B̄, bounds, order, coefs, Lanczos vectors) lives in the FermionDetMatrix's handle."""

    def __init__(self, fermion_det_matrix: FermionDetMatrix, rng=None, rbuf=0.10, n=20, a1=1.0, a2=1.0):
        self.fdm = fermion_det_matrix
        self.handle = fermion_det_matrix.handle
        self.rbuf, self.n = rbuf, n
        self.a1 = 2 * a1 if fermion_det_matrix.is_sym else a1  # :263
        self.a2 = a2
        self.handle.call("smoqy_precond_config", C.c_double(rbuf), int(n), C.c_double(a1), C.c_double(a2))
        self.U = FourierTransformer(np.float64, self.fdm.Lt, self.fdm.N, handle=self.handle)
        self.ϕs = 2 * np.pi / self.fdm.Lt * (np.arange(self.fdm.Lt) + 0.5)  # :220
        update_preconditioner(self, fermion_det_matrix, rng if rng is not None else np.random.default_rng())  # :281

    def _state(self):
        act, norder = C.c_int(0), C.c_int(0)
        bounds = np.zeros(2)
        order = np.zeros(self.fdm.Lt, dtype=np.int32)
        la, lb = np.zeros(self.n), np.zeros(max(self.n - 1, 1))
        self.handle.call("smoqy_precond_get", 0, C.byref(act), bounds.ctypes.data_as(C.POINTER(C.c_double)), order.ctypes.data_as(C.POINTER(C.c_int)), C.byref(norder),
                         la.ctypes.data_as(C.POINTER(C.c_double)), lb.ctypes.data_as(C.POINTER(C.c_double)))
        return bool(act.value), (float(bounds[0]), float(bounds[1])), order[: norder.value].copy(), la, lb[: self.n - 1]

    @property
    def active(self):
        return self._state()[0]

    @property
    def bounds(self):
        return self._state()[1]

    @property
    def order(self):
        return self._state()[2]

    @property
    def coefs(self):
        out = []
        for slot, n in enumerate(self.order):
            c = np.zeros(int(n), dtype=np.complex128)
            self.handle.call("smoqy_precond_get_coefs", 0, slot, L.ptr(c))
            out.append(c.real.copy() if self.fdm.is_sym else c)
        return out


SymKPMPreconditioner = AsymKPMPreconditioner = KPMPreconditioner


def update_preconditioner(P, fermion_det_matrix=None, rng=None, *ignore):
    """update_preconditioner!(Pkpm, fdm, rng), src/KPMPreconditioner.jl:554-597; a no-op for any
    other ``P`` (:600) — this is how ``preconditioner = I`` works."""
    if not isinstance(P, KPMPreconditioner):
        return None
    rng = rng if rng is not None else np.random.default_rng()
    if P.fdm.T is np.complex128:
        # randn!(rng, v) on a Vector{ComplexF64}: (re, im) pairs of variance 1/2 each, :634
        rv = np.ascontiguousarray((rng.standard_normal(2 * P.fdm.N) * np.sqrt(0.5)).view(np.complex128))
    else:
        rv = np.ascontiguousarray(rng.standard_normal(P.fdm.N))  # randn!(rng, v), :634
    P.handle.call("smoqy_precond_update", 0, L.ptr(rv))
    return None


# ---- conjugate gradient -------------------------------------------------------------------------------

def cg_solve(x, A, b, cg_solver: ConjugateGradientSolver | None = None, P=I, maxiter=None, tol=None):
    """cg_solve!(x, A, b, cgs, P; maxiter, tol), src/IterativeSolvers/ConjugateGradient.jl:93-249.
    ``A`` must be a FermionDetMatrix (``mul!(z, A, p)`` ≡ MᵀM, src/FermionDetMatrix.jl:304); the
    loop runs on the device.  ``x is b`` selects the zero initial guess (:112-116).  Returns
    ``(iters, ϵ)``; non-convergence returns ``(maxiter, ϵ)`` without raising."""
    if not isinstance(A, FermionDetMatrix):
        raise TypeError("the device CG solves MᵀM x = b for a FermionDetMatrix A; no generic (CPU) operator path exists")
    cgs = cg_solver or A.cgs
    maxiter = cgs.maxiter if maxiter is None else int(maxiter)
    tol = cgs.tol if tol is None else float(tol)
    if isinstance(P, KPMPreconditioner) and P.handle is not A.handle:
        raise ValueError("the KPMPreconditioner belongs to a different FermionDetMatrix")
    h = A.handle
    xb = L.writable_state(x, h.Lt, h.N)
    same = x is b or (isinstance(b, np.ndarray) and isinstance(x, np.ndarray) and np.shares_memory(x, b))
    bb = xb if same else L.as_state(b, h.Lt, h.N)
    iters = np.zeros(1, dtype=np.int32)
    eps = np.zeros(1)
    h.call("smoqy_cg_solve", L.ptr(xb), L.ptr(bb), int(same), 0, 1, C.c_double(tol), maxiter, int(isinstance(P, KPMPreconditioner)), L.ptr(iters), L.ptr(eps))
    return int(iters[0]), float(eps[0])


# ---- generic mul!/lmul!/ldiv! dispatch -------------------------------------------------------------------

def mul(out, op, v):
    """mul!(v′, fdm, v) ≡ MᵀM (src/FermionDetMatrix.jl:304-313); mul!(u, U, v) ≡ U v, τ → ω
    (src/FourierTransformer.jl:26-36)."""
    if isinstance(op, FermionDetMatrix):
        return mul_MtM(out, op, v)
    if isinstance(op, FourierTransformer):
        o = L.writable_state(out, op.Lτ, op.N)
        o[...] = np.asarray(v).reshape(o.shape, order="F")
        return lmul(op, o)
    raise TypeError(f"mul: unsupported operator {type(op).__name__}")


def lmul(op, v):
    """lmul!(fdm, v) ≡ v = MᵀM v (:292-300); lmul!(U, v) (src/FourierTransformer.jl:39-50)."""
    if isinstance(op, FermionDetMatrix):
        return mul_MtM(v, op, v)
    if isinstance(op, FourierTransformer):
        a = L.writable_state(v, op.Lτ, op.N)
        op.handle.call("smoqy_fft_forward", L.ptr(a), 0, 1)
        return None
    raise TypeError(f"lmul: unsupported operator {type(op).__name__}")


def ldiv(*args, preconditioner=I, rng=None, maxiter=None, tol=None):
    """The reference's ``ldiv!`` methods on the hot path:

    ``ldiv(vp, fdm, v; preconditioner, rng, maxiter, tol)``  v′ = [MᵀM]⁻¹ v   (FermionDetMatrix.jl:248-267)
    ``ldiv(fdm, v; ...)``                                    in place          (:270-288)
    ``ldiv(up, P, u)``                                       u′ = P⁻¹ u        (KPMPreconditioner.jl:355-414, 488-550)
    ``ldiv(U, v)`` / ``ldiv(u, U, v)``                       ω → τ             (FourierTransformer.jl:53-77)
    """
    if len(args) == 3 and isinstance(args[1], FermionDetMatrix):
        vp, fdm, v = args
        update_preconditioner(preconditioner, fdm, rng)  # :259
        return cg_solve(vp, fdm, v, fdm.cgs, preconditioner, maxiter=maxiter, tol=tol)
    if len(args) == 2 and isinstance(args[0], FermionDetMatrix):
        fdm, v = args
        return ldiv(v, fdm, v, preconditioner=preconditioner, rng=rng, maxiter=maxiter, tol=tol)
    if len(args) == 3 and isinstance(args[1], KPMPreconditioner):
        up, P, u = args
        h = P.handle
        if isinstance(up, np.ndarray) and up.dtype == np.float64:
            # the real-vector methods (KPMPreconditioner.jl:288-352, 417-485): the device evaluates the frequencies ω < cld(Lτ, 2),
            # mirrors them as complex conjugates (:334) and returns the real part of the back-transform (:344)
            if not up.flags.f_contiguous or not up.flags.writeable or up.size != h.Lt * h.N:  # strided views (a[::2], a[::-1]) are refused: the library writes Lτ·N contiguous doubles
                raise ValueError("output vector must be a writable float64 Fortran-contiguous (tau-fastest) array of Ltau*N elements")
            uin = np.asfortranarray(np.asarray(u, dtype=np.float64))
            h.call("smoqy_precond_apply_real", L.ptr(up), L.ptr(uin), 0, 1)
            return None
        uin = L.as_state(u, h.Lt, h.N)
        uout = L.writable_state(up, h.Lt, h.N)
        h.call("smoqy_precond_apply", L.ptr(uout), L.ptr(uin), 0, 1)
        return None
    if len(args) == 3 and isinstance(args[1], _Identity):
        up, _, u = args
        L.writable_state(up, np.size(u), 1)[...] = np.asarray(u).reshape(np.shape(up), order="F")
        return None
    if len(args) == 2 and isinstance(args[0], FourierTransformer):
        U, v = args
        a = L.writable_state(v, U.Lτ, U.N)
        U.handle.call("smoqy_fft_inverse", L.ptr(a), 0, 1)
        return None
    if len(args) == 3 and isinstance(args[1], FourierTransformer):
        u, U, v = args
        o = L.writable_state(u, U.Lτ, U.N)
        o[...] = np.asarray(v).reshape(o.shape, order="F")
        return ldiv(U, o)
    raise TypeError("ldiv: unsupported argument combination")


# ---- Holstein shift matrix Λ ------------------------------------------------------------------------------

def update_Λ(Λ, electron_phonon_parameters: ElectronPhononParameters, handle: L.Handle | None = None):
    """update_Λ!(Λ, elph), src/holstein_shift_matrix.jl:2-44, computed on the device and written
    into ``Λ`` (``Ltau x N``)."""
    elph = electron_phonon_parameters
    Lt, N = Λ.shape
    if handle is None:
        handle = L.Handle(Lt, N, np.zeros((2, 0), dtype=np.int64), np.zeros((2, 0), dtype=np.int64), True, 1, 1, -1)
    _lambda_update_device(handle, elph)
    out = Λ if (Λ.flags.f_contiguous and Λ.dtype == np.float64) else np.zeros((Lt, N), order="F")
    handle.call("smoqy_lambda_get", 0, L.ptr(out))
    if out is not Λ:
        Λ[...] = out


def _lambda_update_device(handle, elph):
    hol = elph.holstein
    x = np.asfortranarray(elph.x, dtype=np.float64)
    if hol is None:
        handle.call("smoqy_lambda_update", 0, L.ptr(x), x.shape[0], C.c_double(elph.dtau), 0, None, None, None, None, None)
        return
    handle.call("smoqy_lambda_update", 0, L.ptr(x), x.shape[0], C.c_double(elph.dtau), len(hol.alpha), L.ptr(np.ascontiguousarray(hol.coupling_to_phonon, dtype=np.int64)),
                L.ptr(np.ascontiguousarray(hol.coupling_to_site, dtype=np.int64)), L.ptr(np.ascontiguousarray(hol.alpha, dtype=np.float64)), L.ptr(np.ascontiguousarray(hol.alpha3, dtype=np.float64)),
                L.ptr(np.ascontiguousarray(hol.ph_sym_form, dtype=np.int32)))


_scratch_handles: dict = {}


def _lambda_apply(op, up, Λ, u, handle=None):
    Lt, N = Λ.shape
    if handle is None:
        key = (Lt, N)
        if key not in _scratch_handles:
            _scratch_handles[key] = L.Handle(Lt, N, np.zeros((2, 0), dtype=np.int64), np.zeros((2, 0), dtype=np.int64), True, 1, 1, -1)
        handle = _scratch_handles[key]
    lam = np.asfortranarray(Λ, dtype=np.float64)
    uin = L.as_state(u, Lt, N)
    uout = L.writable_state(up, Lt, N)
    handle.call("smoqy_lambda_apply", op, L.ptr(uout), L.ptr(uin), L.ptr(lam), 0, 1)


def mul_Λ(up, Λ, u, handle=None):
    """mul_Λ!: |u′⟩ = Λ|u⟩ (src/holstein_shift_matrix.jl:47-71); ``up is u`` allowed."""
    _lambda_apply(L.LAMBDA_MUL, up, Λ, u, handle)


def ldiv_Λ(up, Λ, u, handle=None):
    """ldiv_Λ!: |u′⟩ = Λ⁻¹|u⟩ (:74-98)."""
    _lambda_apply(L.LAMBDA_LDIV, up, Λ, u, handle)


def mul_Λᵀ(up, Λ, u, handle=None):
    """mul_Λᵀ!: |u′⟩ = Λᵀ|u⟩ (:102-126)."""
    _lambda_apply(L.LAMBDA_MULT, up, Λ, u, handle)


def ldiv_Λᵀ(up, Λ, u, handle=None):
    """ldiv_Λᵀ!: |u′⟩ = Λ⁻ᵀ|u⟩ (:129-153)."""
    _lambda_apply(L.LAMBDA_LDIVT, up, Λ, u, handle)


update_Lambda, mul_Lambda, ldiv_Lambda, mul_LambdaT, ldiv_LambdaT = update_Λ, mul_Λ, ldiv_Λ, mul_Λᵀ, ldiv_Λᵀ


# ---- PFFCalculator ---------------------------------------------------------------------------------------

class PFFCalculator:
    """src/PFFCalculator.jl:9-53.  Φ, Λ, u, u′, u″ are device-resident (vectors of the
    FermionDetMatrix's handle); the properties download a host copy on access."""

    def __init__(self, electron_phonon_parameters: ElectronPhononParameters, fermion_det_matrix: FermionDetMatrix):
        self.fdm = fermion_det_matrix
        self.handle = fermion_det_matrix.handle
        if fermion_det_matrix.nrhs != 1:
            raise ValueError("PFFCalculator expects a FermionDetMatrix with nrhs = 1")
        self._phi, self._u, self._u1, self._u2 = (self.handle.vec_alloc() for _ in range(4))

    @property
    def Φ(self):
        return self.handle.vec_download(self._phi)

    @property
    def u(self):
        return self.handle.vec_download(self._u)

    @property
    def Λ(self):
        out = np.zeros((self.fdm.Lt, self.fdm.N), order="F")
        self.handle.call("smoqy_lambda_get", 0, L.ptr(out))
        return out


def _complex_randn(rng, Lt, N):
    """randn!(rng, Φ) for a ComplexF64 matrix: (re, im) pairs in memory order, variance 1/2 each."""
    flat = rng.standard_normal(2 * Lt * N) * np.sqrt(0.5)
    return flat.view(np.complex128).reshape((Lt, N), order="F")


def sample_pseudofermion_fields(pff_calculator: PFFCalculator, electron_phonon_parameters, fermion_det_matrix, rng=None, R=None):
    """sample_pseudofermion_fields!: Φ = Λᵀ Mᵀ R, returns |R|² (src/PFFCalculator.jl:56-76).
    ``R`` may be supplied instead of being drawn from ``rng`` (testing aid)."""
    pff, h = pff_calculator, pff_calculator.handle
    _lambda_update_device(h, electron_phonon_parameters)  # update_Λ!  :63
    if R is None:
        rng = rng if rng is not None else np.random.default_rng()
        R = _complex_randn(rng, h.Lt, h.N)                # randn!     :67
    h.vec_upload(pff._phi, R)
    Sf = float(h.vec_dot(pff._phi, pff._phi)[0].real)     # |R|²       :69
    h.call("smoqy_matvec_v", L.OP_MT, pff._phi, pff._phi)  # lmul_Mt!   :71
    h.call("smoqy_lambda_apply_v", L.LAMBDA_MULT, pff._phi, pff._phi)  # mul_Λᵀ! :73
    return Sf


def calculate_fermionic_action(pff_calculator: PFFCalculator, electron_phonon_parameters, fermion_det_matrix, preconditioner=I, rng=None, tol=None, maxiter=None):
    """calculate_fermionic_action!: S_f = Φᵀ Λ⁻¹ [MᵀM]⁻¹ Λ⁻ᵀ Φ; returns ``(Sf, iters, ϵ)``
    (src/PFFCalculator.jl:79-116).  A complex action beyond √tol warns like the reference (:110-112)."""
    pff, h, fdm = pff_calculator, pff_calculator.handle, fermion_det_matrix
    tol = fdm.cgs.tol if tol is None else float(tol)
    maxiter = fdm.cgs.maxiter if maxiter is None else int(maxiter)
    _lambda_update_device(h, electron_phonon_parameters)                  # update_Λ!  :94
    h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIVT, pff._u, pff._phi)      # Ψ = Λ⁻ᵀ Φ  :97
    update_preconditioner(preconditioner, fdm, rng)                       # FermionDetMatrix.jl:259
    iters = np.zeros(1, dtype=np.int32)
    eps = np.zeros(1)
    h.call("smoqy_cg_solve_v", pff._u, pff._u, C.c_double(tol), maxiter, int(isinstance(preconditioner, KPMPreconditioner)), L.ptr(iters), L.ptr(eps))  # :99-105
    h.call("smoqy_lambda_apply_v", L.LAMBDA_LDIV, pff._u, pff._u)         # Ψ = Λ⁻¹ Ψ  :107
    Sf = complex(h.vec_dot(pff._phi, pff._u)[0])                           # Φ·Ψ        :109
    if np.sqrt(tol) < abs(Sf.imag / Sf.real):
        warnings.warn(f"Complex Fermionic Action Encountered. Sf={Sf} tol={tol}")
    return Sf.real, int(iters[0]), float(eps[0])


# ---- force terms -------------------------------------------------------------------

def set_force_couplings(fermion_det_matrix: FermionDetMatrix, couplings):
    """Hand the flattened electron-phonon couplings (``lattice.ForceCouplings``) to the device once;
    what the reference reads from ``ElectronPhononParameters`` inside the force routines."""
    s, keep = L.couplings_struct(couplings)
    fermion_det_matrix.handle.call("smoqy_force_set_couplings", C.byref(s))
    fermion_det_matrix._force_couplings = couplings
    set_force_phonons(fermion_det_matrix, couplings.x)


def set_force_phonons(fermion_det_matrix: FermionDetMatrix, x):
    """Refresh the device copy of the phonon fields ``x`` (Nph x Ltau) after they moved."""
    xx = np.asfortranarray(x, dtype=np.float64)
    fermion_det_matrix.handle.call("smoqy_force_set_phonons", L.ptr(xx))


def set_bare_model(fermion_det_matrix: FermionDetMatrix, V0, t0, perm):
    """Bare on-site energies ``V0`` (N) and hoppings ``t0`` (Nh, FermionPathIntegral order): what
    ``SmoQyDQMC.update!(fermion_path_integral, elph, x, -1)`` leaves in the path integral
    (src/EFAPFFHMCUpdater.jl:148, 200).  Needed once before ``update_from_phonons``."""
    h = fermion_det_matrix.handle
    h.call("smoqy_set_bare_model", L.ptr(np.ascontiguousarray(V0, dtype=np.float64)), L.ptr(np.ascontiguousarray(t0, dtype=np.float64)), L.ptr(np.ascontiguousarray(perm, dtype=np.int64)))


def update_from_phonons(fermion_det_matrix: FermionDetMatrix, x):
    """``update!(fermion_path_integral, elph, x, +1); update!(fermion_det_matrix, fermion_path_integral);
    update_Λ!`` in one device pass from the phonon fields ``x`` (Nph x Ltau), as at
    src/EFAPFFHMCUpdater.jl:200-205 — the host never forms V, t or Λ."""
    xx = np.asfortranarray(x, dtype=np.float64)
    fermion_det_matrix.handle.call("smoqy_update_from_phonons_all", L.ptr(xx))


def _tmp_vecs(fdm, n):
    if not hasattr(fdm, "_tmp_ids"):
        fdm._tmp_ids = []
    while len(fdm._tmp_ids) < n:
        fdm._tmp_ids.append(fdm.handle.vec_alloc())
    return fdm._tmp_ids[:n]


def mul_nuRe_dMdx(νRedMdx, ν, u, v, fermion_det_matrix, elph=None):
    """mul_νRe∂M∂x! (``∂`` is not a Python identifier character, hence the ASCII name): accumulates ν·Re⟨u|∂M/∂x|v⟩ into the
    ``Nph x Ltau`` array (src/fermion_det_matrix_dervative.jl:2-186).  The couplings must have been
    set with ``set_force_couplings``."""
    fdm, h = fermion_det_matrix, fermion_det_matrix.handle
    iu, iv = _tmp_vecs(fdm, 2)
    h.vec_upload(iu, L.as_state(u, h.Lt, h.N))
    h.vec_upload(iv, L.as_state(v, h.Lt, h.N))
    out = νRedMdx if (νRedMdx.flags.f_contiguous and νRedMdx.dtype == np.float64) else np.asfortranarray(νRedMdx, dtype=np.float64)
    h.call("smoqy_force_dMdx_v", C.c_double(ν), iu, iv, L.ptr(out))
    if out is not νRedMdx:
        νRedMdx[...] = out


def mul_nuRe_dLdx(νRedΛdx, ν, up, u, Λ, fermion_det_matrix):
    """mul_νRe∂Λ∂x!(νRe∂Λ∂x, ν, u′, u, Λ, elph) (src/holstein_shift_matrix.jl:156-201)."""
    fdm, h = fermion_det_matrix, fermion_det_matrix.handle
    iu, iv = _tmp_vecs(fdm, 2)
    h.vec_upload(iu, L.as_state(up, h.Lt, h.N))
    h.vec_upload(iv, L.as_state(u, h.Lt, h.N))
    lam = np.asfortranarray(Λ, dtype=np.float64)
    h.call("smoqy_lambda_set", 0, L.ptr(lam))
    out = νRedΛdx if (νRedΛdx.flags.f_contiguous and νRedΛdx.dtype == np.float64) else np.asfortranarray(νRedΛdx, dtype=np.float64)
    h.call("smoqy_force_dLdx_v", C.c_double(ν), iu, iv, L.ptr(out))
    if out is not νRedΛdx:
        νRedΛdx[...] = out



def calculate_derivative_fermionic_action(dSfdx, pff_calculator: PFFCalculator, electron_phonon_parameters, fermion_det_matrix, preconditioner=I, rng=None, tol=None, maxiter=None):
    """calculate_derivative_fermionic_action! (src/PFFCalculator.jl:119-157): returns
    ``(Sf, iters, ϵ)`` and accumulates ∂S_f/∂x into ``dSfdx`` (``Nph x Ltau``)."""
    Sf, iters, eps = calculate_fermionic_action(pff_calculator, electron_phonon_parameters, fermion_det_matrix, preconditioner, rng, tol, maxiter)
    set_force_phonons(fermion_det_matrix, fermion_det_matrix._force_couplings.x)
    out = dSfdx if (dSfdx.flags.f_contiguous and dSfdx.dtype == np.float64) else np.asfortranarray(dSfdx, dtype=np.float64)
    pff_calculator.handle.call("smoqy_force_v", pff_calculator._u, L.ptr(out))
    if out is not dSfdx:
        dSfdx[...] = out
    return Sf, iters, eps


# ---- GreensEstimator ------------------------------------------------------------

class GreensEstimator:
    """GreensEstimator(fermion_det_matrix, model_geometry; Nrv, preconditioner, rng, maxiter, tol),
    src/Measurements/GreensEstimator.jl:10-121.  ``model_geometry`` is anything with ``unit_cell.n`` and
    ``lattice.L``, or the pair ``(n, L)``.  The ``Nrv`` solves of ``update_greens_estimator!`` run as ONE batched
    solve on a second handle created with ``nrhs = Nrv`` that follows the sampling handle's fields device to device;
    ``GR`` and ``Rt`` (shape ``(Lτ, n, L..., Nrv)``, the reference's arrays) come back to the host for the contractions
    that stay there, while ``measure_GΔ0`` contracts them on the device."""

    def __init__(self, fermion_det_matrix: FermionDetMatrix, model_geometry, Nrv=10, preconditioner=I, rng=None, maxiter=None, tol=None):
        fdm = fermion_det_matrix
        if hasattr(model_geometry, "unit_cell"):
            n, Ls = int(model_geometry.unit_cell.n), tuple(int(x) for x in model_geometry.lattice.L)
        else:
            n, Ls = int(model_geometry[0]), tuple(int(x) for x in model_geometry[1])
        self.Nrv, self.n, self.L = int(Nrv), n, Ls
        self.N = int(np.prod(Ls))                     # unit cells (:84)
        self.V = fdm.Lt * fdm.N                       # :86
        self.Lτ = self.V // (n * self.N)              # :88
        if n * self.N != fdm.N:
            raise ValueError("unit cell / lattice do not match the FermionDetMatrix")
        self.handle = L.Handle(fdm.Lt, fdm.N, fdm.checkerboard_neighbor_table, fdm._colors, fdm.is_sym, 1, self.Nrv, fdm.handle.device)
        self.handle.call("smoqy_ge_config", n, len(Ls), L.ptr(np.asarray(Ls, dtype=np.int64)))
        self._r, self._gr, self._mtr = (self.handle.vec_alloc() for _ in range(3))   # Rt, GR, MtR (:91-93); GR starts at zero
        shape = (self.Lτ, n) + Ls + (self.Nrv,)
        self.Rt = np.zeros(shape, dtype=np.complex128, order="F")
        self.GR = np.zeros(shape, dtype=np.complex128, order="F")
        self._pre_cfg = None
        update_greens_estimator(self, fdm, preconditioner=preconditioner, rng=rng, maxiter=maxiter, tol=tol)  # :111-118

    @property
    def CΔ0_shape(self):
        return (self.Lτ + 1,) + self.L


def update_greens_estimator(greens_estimator: GreensEstimator, fermion_det_matrix: FermionDetMatrix, preconditioner=I, rng=None, maxiter=None, tol=None):
    """update_greens_estimator!(ge, fdm; preconditioner, rng, maxiter, tol), src/Measurements/GreensEstimator.jl:125-175.
    Returns the average iteration count.  All ``Nrv`` right-hand sides advance in one batched CG."""
    ge, fdm, h = greens_estimator, fermion_det_matrix, greens_estimator.handle
    rng = rng if rng is not None else np.random.default_rng()
    maxiter = fdm.cgs.maxiter if maxiter is None else int(maxiter)
    tol = fdm.cgs.tol if tol is None else float(tol)
    h.call("smoqy_copy_fields", 0, fdm.handle._h, 0)
    # randn!(rng, R); R ./= abs.(R)  (:141-142), one V-vector per random vector, in the reference's memory order
    R = np.empty((fdm.Lt, fdm.N, ge.Nrv), dtype=np.complex128, order="F")
    flat = R.reshape(-1, order="F").view(np.float64)
    rng.standard_normal(out=flat)
    R /= np.abs(R)
    use_pre = isinstance(preconditioner, KPMPreconditioner)
    if use_pre:  # update_preconditioner!(preconditioner, fdm, rng)  (:150), on the measurement handle
        cfg = (preconditioner.rbuf, preconditioner.n, preconditioner.a1 / (2 if fdm.is_sym else 1), preconditioner.a2)
        if ge._pre_cfg != cfg:
            h.call("smoqy_precond_config", C.c_double(cfg[0]), int(cfg[1]), C.c_double(cfg[2]), C.c_double(cfg[3]))
            ge._pre_cfg = cfg
        rv = np.ascontiguousarray(rng.standard_normal(fdm.N))
        h.call("smoqy_precond_update", 0, L.ptr(rv))
    h.vec_upload(ge._r, R)
    h.call("smoqy_matvec_v", L.OP_MT, ge._mtr, ge._r)                    # mul_Mt!(MtR, fdm, R′)   :157
    iters = np.zeros(ge.Nrv, dtype=np.int32)
    eps = np.zeros(ge.Nrv)
    h.call("smoqy_cg_solve_v", ge._gr, ge._mtr, C.c_double(tol), maxiter, int(use_pre), L.ptr(iters), L.ptr(eps))  # ldiv!(GR′, fdm, MtR)  :159-165 (GR′ is the initial guess)
    shape = ge.GR.shape
    ge.GR[...] = h.vec_download(ge._gr).reshape(shape, order="F")
    ge.Rt[...] = np.conj(R).reshape(shape, order="F")                    # :171
    return float(iters.mean())


def measure_GΔ0(correlation, greens_estimator: GreensEstimator, orbitals):
    """measure_GΔ0!(correlation, ge, (a, b)), src/Measurements/GreensEstimator.jl:179-233: adds the translation-averaged
    G(Δ,0) (shape ``L... x (Lτ+1)``) to ``correlation``; the cross-correlations run on the device."""
    ge, h = greens_estimator, greens_estimator.handle
    a, b = (int(x) for x in orbitals)
    out = np.zeros(ge.CΔ0_shape, dtype=np.complex128, order="F")
    h.call("smoqy_ge_measure_GD0", ge._gr, ge._r, a, b, L.ptr(out))
    correlation += np.moveaxis(out, 0, -1)                               # add_contraction_to_correlation!  :712-726
    return None


measure_GD0 = measure_GΔ0


# four-point estimators (src/Measurements/GreensEstimator.jl:241-606): the pair sums and the scalar boundary terms are reduced on
# the device from the resident GR and R; the mirror only decides which element of CΔ0 each boundary term goes to

def _ge_pair_sum(ge: GreensEstimator, slots, tΔ, t0, conj_tΔ, conj_t0):
    h = ge.handle
    arr = (L.GeSlot * 4)()
    for q, (source, orbital, shift, second) in enumerate(slots):
        arr[q].source, arr[q].orbital, arr[q].second = int(source), int(orbital), int(second)
        for d in range(2):
            arr[q].shift[d] = int(shift[d]) if d < len(shift) else 0
    wshape = (ge.Lτ,) + ge.L
    keep = []

    def weights(t):
        if t is None:
            return None
        w = np.asfortranarray(np.broadcast_to(np.asarray(t), wshape), dtype=np.complex128)
        keep.append(w)
        return L.ptr(w)

    out = np.zeros(ge.CΔ0_shape, dtype=np.complex128, order="F")
    h.call("smoqy_ge_measure_pairs", ge._gr, ge._r, C.cast(arr, C.c_void_p), weights(tΔ), int(bool(conj_tΔ)), weights(t0), int(bool(conj_t0)), L.ptr(out))
    return out


def _bconj(x, flag):
    return np.conj(x) if flag else x


def _mod1(x, Ln):
    return (int(x) - 1) % int(Ln)


def _boundary_dot(ge, orbital_gr, orbital_r, shift, tΔ, t0, conj_tΔ, conj_t0, tshift):
    """Σ_rv Σ_i [bconj(tβ)·bconj(t0)]·circshift(GR_o, (0, shift...))[i]·Rt_o′[i] / (Nrv·length) — the scalar of :325-334 and its
    siblings, reduced on the device from the resident GR and R."""
    h = ge.handle
    sh = np.zeros(2, dtype=np.int64)
    ts = np.zeros(2, dtype=np.int64)
    sh[: len(shift)] = shift
    ts[: len(tshift)] = tshift
    wshape = (ge.Lτ,) + ge.L
    wD = w0 = None
    if tΔ is not None or t0 is not None:
        wD = np.asfortranarray(np.broadcast_to(np.asarray(1.0 if tΔ is None else tΔ), wshape), dtype=np.complex128)
        w0 = np.asfortranarray(np.broadcast_to(np.asarray(1.0 if t0 is None else t0), wshape), dtype=np.complex128)
    out = np.zeros(1, dtype=np.complex128)
    h.call("smoqy_ge_boundary_dot", ge._gr, ge._r, int(orbital_gr), int(orbital_r), L.ptr(sh), None if wD is None else L.ptr(wD), int(bool(conj_tΔ)), L.ptr(ts),
           None if w0 is None else L.ptr(w0), int(bool(conj_t0)), L.ptr(out))
    return out[0]


def measure_GΔ0_GΔ0(correlation, greens_estimator: GreensEstimator, orbitals, r1, r2, r3, r4, coef, tΔ=None, t0=None, conj_tΔ=False, conj_t0=False):
    """measure_GΔ0_GΔ0! (src/Measurements/GreensEstimator.jl:241-388): (GR_a^{r1} ⊙ GR_c^{r3}) ⋆ (Rt_b^{r2} ⊙ Rt_d^{r4}) over all pairs of
    random vectors, plus the τ = β boundary terms."""
    ge = greens_estimator
    a, b, c, d = (int(x) for x in orbitals)
    D, Ls, Lt = len(ge.L), ge.L, ge.Lτ
    G = _ge_pair_sum(ge, [(0, a, r1, 0), (0, c, r3, 1), (1, b, r2, 0), (1, d, r4, 1)], tΔ, t0, conj_tΔ, conj_t0)  # :285-306
    if a == b:   # :312-337
        idx = (Lt,) + tuple(_mod1(1 - r1[k] + r2[k], Ls[k]) for k in range(D))
        G[idx] -= _boundary_dot(ge, c, d, [r1[k] - r2[k] - r3[k] + r4[k] for k in range(D)], tΔ, t0, conj_tΔ, conj_t0, [r1[k] - r2[k] for k in range(D)])
    if c == d:   # :341-364
        idx = (Lt,) + tuple(_mod1(1 - r3[k] + r4[k], Ls[k]) for k in range(D))
        G[idx] -= _boundary_dot(ge, a, b, [-r1[k] + r2[k] + r3[k] - r4[k] for k in range(D)], tΔ, t0, conj_tΔ, conj_t0, [r3[k] - r4[k] for k in range(D)])
    if a == b and c == d and all((r2[k] - r1[k]) % Ls[k] == (r4[k] - r3[k]) % Ls[k] for k in range(D)):   # :367-382
        idx = (Lt,) + tuple(_mod1(1 + r2[k] - r1[k], Ls[k]) for k in range(D))
        if tΔ is None and t0 is None:
            G[idx] += 1
        else:
            tb = np.roll(np.asarray(tΔ), shift=tuple(int(r1[k] - r2[k]) for k in range(D)), axis=tuple(range(1, 1 + D)))
            G[idx] += np.sum(_bconj(tb, conj_tΔ) * _bconj(np.asarray(t0), conj_t0)) / tb.size
    correlation += coef * np.moveaxis(G, 0, -1)   # :385
    return None


def measure_GΔΔ_G00(correlation, greens_estimator: GreensEstimator, orbitals, r1, r2, r3, r4, coef, tΔ=None, t0=None, conj_tΔ=False, conj_t0=False):
    """measure_GΔΔ_G00! (:396-467): (GR_a^{r1} ⊙ Rt_b^{r2}) ⋆ (GR_c^{r3} ⊙ Rt_d^{r4}); no boundary terms."""
    a, b, c, d = (int(x) for x in orbitals)
    G = _ge_pair_sum(greens_estimator, [(0, a, r1, 0), (1, b, r2, 0), (0, c, r3, 1), (1, d, r4, 1)], tΔ, t0, conj_tΔ, conj_t0)
    correlation += coef * np.moveaxis(G, 0, -1)
    return None


def measure_G0Δ_GΔ0(correlation, greens_estimator: GreensEstimator, orbitals, r1, r2, r3, r4, coef, tΔ=None, t0=None, conj_tΔ=False, conj_t0=False):
    """measure_G0Δ_GΔ0! (:475-606): (Rt_b^{r2} ⊙ GR_c^{r3}) ⋆ (GR_a^{r1} ⊙ Rt_d^{r4}), boundary terms at τ = 0 and τ = β."""
    ge = greens_estimator
    a, b, c, d = (int(x) for x in orbitals)
    D, Ls, Lt = len(ge.L), ge.L, ge.Lτ
    G = _ge_pair_sum(ge, [(1, b, r2, 0), (0, c, r3, 1), (0, a, r1, 0), (1, d, r4, 1)], tΔ, t0, conj_tΔ, conj_t0)  # :518-539
    sh = [-r1[k] + r2[k] - r3[k] + r4[k] for k in range(D)]
    if a == b:   # :545-569, τ = 0
        idx = (0,) + tuple(_mod1(1 + r1[k] - r2[k], Ls[k]) for k in range(D))
        G[idx] -= _boundary_dot(ge, c, d, sh, tΔ, t0, conj_tΔ, conj_t0, [-r1[k] + r2[k] for k in range(D)])
    if c == d:   # :575-599, τ = β
        idx = (Lt,) + tuple(_mod1(1 + r4[k] - r3[k], Ls[k]) for k in range(D))
        G[idx] -= _boundary_dot(ge, a, b, sh, tΔ, t0, conj_tΔ, conj_t0, [-r4[k] + r3[k] for k in range(D)])
    correlation += coef * np.moveaxis(G, 0, -1)   # :603
    return None


measure_GD0_GD0, measure_GDD_G00, measure_G0D_GD0 = measure_GΔ0_GΔ0, measure_GΔΔ_G00, measure_G0Δ_GΔ0
