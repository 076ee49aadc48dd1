"""ctypes front-end of ``smoqy_oracle.c`` (CPU restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY — see ``oracle/__init__.py``.  All arrays use the reference layout:
state vectors / fields are ``(Ltau, N)`` Fortran-ordered numpy arrays (tau contiguous), the
neighbour table is ``(2, Nh)`` Fortran-ordered int64, 1-based and colour sorted.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMOQY_ORACLE_LIB selects another build of the same source (the sanitizer build of tests/test_oracle_sanitizers.py)
_LIB_PATH = os.environ.get("SMOQY_ORACLE_LIB") or os.path.join(_HERE, "libsmoqy_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (a few seconds)."""
    src = os.path.join(_HERE, "smoqy_oracle.c")
    if os.environ.get("SMOQY_ORACLE_LIB"):
        return _LIB_PATH  # built by whoever set the variable
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libsmoqy_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fdm_create.restype = C.c_void_p
        _lib.orc_fdm_create_c.restype = C.c_void_p
        _lib.orc_kpm_create.restype = C.c_void_p
        _lib.orc_fft_create.restype = C.c_void_p
        _lib.orc_cg_solve.restype = C.c_int
        _lib.orc_kpm_active.restype = C.c_int
        _lib.orc_kpm_order.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fvec(a, dtype=np.complex128):
    """Return a Fortran-ordered, owned copy of ``a`` with the given dtype."""
    return np.array(a, dtype=dtype, order="F", copy=True)


class OracleFDM:
    """Restatement of Sym/AsymFermionDetMatrix (src/FermionDetMatrix.jl:44-55, 137-148)."""

    def __init__(self, neighbor_table, expV, cosh, sinh, is_sym=True):
        self.nt = np.asfortranarray(neighbor_table, dtype=np.int64)
        self.expV = np.asfortranarray(expV, dtype=np.float64)
        self.cosh = np.asfortranarray(np.real(cosh), dtype=np.float64)
        # T = ComplexF64 (complex hoppings): sinhΔτt carries the phase sign(conj t); kept as separate real / imaginary arrays
        self.is_complex = bool(np.iscomplexobj(sinh))
        self.sinh = np.asfortranarray(np.real(sinh), dtype=np.float64)
        self.sinh_im = np.asfortranarray(np.imag(sinh), dtype=np.float64) if self.is_complex else None
        self.Lt, self.N = self.expV.shape
        self.Nh = self.nt.shape[1]
        self.is_sym = bool(is_sym)
        assert self.cosh.shape == (self.Lt, self.Nh) and self.sinh.shape == (self.Lt, self.Nh)
        self._h = C.c_void_p(lib().orc_fdm_create_c(self.Lt, self.N, self.Nh, int(self.is_sym), _p(self.nt), _p(self.expV), _p(self.cosh), _p(self.sinh),
                                                    _p(self.sinh_im) if self.is_complex else None))

    def __del__(self):
        try:
            lib().orc_fdm_destroy(self._h)
        except Exception:
            pass

    def _apply(self, fn, v):
        v = fvec(v).reshape(self.Lt, self.N, order="F")
        out = np.zeros_like(v, order="F")
        fn(self._h, _p(out), _p(v))
        return out

    def mul_M(self, v):
        return self._apply(lib().orc_mul_M, v)

    def mul_Mt(self, v):
        return self._apply(lib().orc_mul_Mt, v)

    def mul_MtM(self, v):
        return self._apply(lib().orc_mul_MtM, v)

    def mul_MMt(self, v):
        return self._apply(lib().orc_mul_MMt, v)

    def checkerboard(self, v, transposed=False, inverse=False, interval=None):
        v = fvec(v).reshape(self.Lt, self.N, order="F")
        h0, h1 = (0, self.Nh) if interval is None else interval
        fn = lib().orc_checkerboard_ldiv_c if inverse else lib().orc_checkerboard_lmul_c
        fn(_p(v), self.Lt, self.N, _p(self.nt), _p(self.cosh), _p(self.sinh), _p(self.sinh_im) if self.is_complex else None, int(transposed), int(h0), int(h1))
        return v

    def cg_solve(self, b, x0=None, precond=None, tol=1e-10, maxiter=10000):
        """cg_solve! (src/IterativeSolvers/ConjugateGradient.jl:93-249).  ``x0=None`` is the
        ``x === b`` case (zero initial guess).  Returns (x, iters, eps)."""
        b = fvec(b).reshape(self.Lt, self.N, order="F")
        x = np.zeros_like(b, order="F") if x0 is None else fvec(x0).reshape(self.Lt, self.N, order="F")
        eps = C.c_double(0.0)
        ph = precond._h if precond is not None else None
        it = lib().orc_cg_solve(self._h, ph, _p(x), _p(b), int(x0 is None), C.c_double(tol), int(maxiter), C.byref(eps))
        return x, int(it), float(eps.value)


def update_fields(V, t, perm, dtau, is_sym=True):
    """update!(fdm, fpi) (src/FermionDetMatrix.jl:208-236).  V is (N, Ltau), t is (Nh, Ltau),
    perm is the 1-based checkerboard permutation.  Returns (expV, cosh, sinh)."""
    V = np.asfortranarray(V, dtype=np.float64)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    N, Lt = V.shape
    Nh = np.shape(t)[0]
    expV = np.zeros((Lt, N), order="F")
    ch = np.zeros((Lt, Nh), order="F")
    sh = np.zeros((Lt, Nh), order="F")
    if np.iscomplexobj(t):  # T = ComplexF64: returns a complex sinh array
        tr, ti = np.asfortranarray(np.real(t), dtype=np.float64), np.asfortranarray(np.imag(t), dtype=np.float64)
        shi = np.zeros((Lt, Nh), order="F")
        lib().orc_update_fields_c(_p(expV), _p(ch), _p(sh), _p(shi), Lt, N, Nh, _p(V), _p(tr), _p(ti), _p(perm), C.c_double(dtau), int(is_sym))
        return expV, ch, np.asfortranarray(sh + 1j * shi)
    t = np.asfortranarray(t, dtype=np.float64)
    lib().orc_update_fields(_p(expV), _p(ch), _p(sh), Lt, N, Nh, _p(V), _p(t), _p(perm), C.c_double(dtau), int(is_sym))
    return expV, ch, sh


LAMBDA_OPS = {"mul": 0, "ldiv": 1, "mulT": 2, "ldivT": 3}


def lambda_apply(Lam, v, op):
    """mul_Λ!, ldiv_Λ!, mul_Λᵀ!, ldiv_Λᵀ! (src/holstein_shift_matrix.jl:47-153)."""
    Lam = np.asfortranarray(Lam, dtype=np.float64)
    Lt, N = Lam.shape
    v = fvec(v).reshape(Lt, N, order="F")
    out = np.zeros_like(v, order="F")
    lib().orc_lambda_apply(_p(out), _p(v), _p(Lam), Lt, N, LAMBDA_OPS[op])
    return out


def update_lambda(Lt, N, x, dtau, coupling_to_phonon, coupling_to_site, alpha, alpha3, ph_sym):
    """update_Λ! (src/holstein_shift_matrix.jl:2-44); x is (Nph, Ltau); ids are 1-based."""
    x = np.asfortranarray(x, dtype=np.float64)
    Lam = np.zeros((Lt, N), order="F")
    c2p = np.ascontiguousarray(coupling_to_phonon, dtype=np.int64)
    c2s = np.ascontiguousarray(coupling_to_site, dtype=np.int64)
    al = np.ascontiguousarray(alpha, dtype=np.float64)
    a3 = np.ascontiguousarray(alpha3, dtype=np.float64)
    ps = np.ascontiguousarray(ph_sym, dtype=np.int32)
    lib().orc_update_lambda(_p(Lam), Lt, N, _p(x), x.shape[0], C.c_double(dtau), len(c2p), _p(c2p), _p(c2s), _p(al), _p(a3), _p(ps))
    return Lam


class OracleFT:
    """FourierTransformer (src/FourierTransformer.jl)."""

    def __init__(self, Lt, N):
        self.Lt, self.N = Lt, N
        self._h = C.c_void_p(lib().orc_fft_create(Lt))

    def __del__(self):
        try:
            lib().orc_fft_destroy(self._h)
        except Exception:
            pass

    def forward(self, v):
        v = fvec(v).reshape(self.Lt, self.N, order="F")
        lib().orc_ft_forward(self._h, _p(v), self.Lt, self.N)
        return v

    def inverse(self, v):
        v = fvec(v).reshape(self.Lt, self.N, order="F")
        lib().orc_ft_inverse(self._h, _p(v), self.Lt, self.N)
        return v


class OracleKPM:
    """Sym/AsymKPMPreconditioner (src/KPMPreconditioner.jl:61-99, 132-170, 198-284)."""

    def __init__(self, fdm: OracleFDM, rbuf=0.10, n=20, a1=1.0, a2=1.0):
        self.fdm = fdm
        self.n = n
        self._h = C.c_void_p(lib().orc_kpm_create(fdm.Lt, fdm.N, fdm.Nh, int(fdm.is_sym), _p(fdm.nt), C.c_double(rbuf), int(n), C.c_double(a1), C.c_double(a2)))

    def __del__(self):
        try:
            lib().orc_kpm_destroy(self._h)
        except Exception:
            pass

    def update(self, randvec):
        """update_preconditioner! (:554-597); ``randvec`` = the N normal deviates drawn at :634."""
        f = self.fdm
        if f.is_complex:  # randn!(rng, v) on a Vector{ComplexF64}: N complex deviates
            rv = np.ascontiguousarray(randvec, dtype=np.complex128)
            assert rv.shape == (f.N,)
            rr, ri = np.ascontiguousarray(rv.real), np.ascontiguousarray(rv.imag)
            lib().orc_kpm_update_c(self._h, _p(f.expV), _p(f.cosh), _p(f.sinh), _p(f.sinh_im), _p(rr), _p(ri))
            return
        rv = np.ascontiguousarray(randvec, dtype=np.float64)
        assert rv.shape == (f.N,)
        lib().orc_kpm_update(self._h, _p(f.expV), _p(f.cosh), _p(f.sinh), _p(rv))

    def apply(self, v):
        """ldiv!(u', P, u), complex method (:355-414 / :488-550)."""
        f = self.fdm
        v = fvec(v).reshape(f.Lt, f.N, order="F")
        out = np.zeros_like(v, order="F")
        lib().orc_kpm_apply(self._h, _p(out), _p(v))
        return out

    def apply_real(self, v):
        """ldiv!(u', P, u), real-vector method (:288-352 / :417-485): half the frequencies, conjugate mirror, real part."""
        f = self.fdm
        v = np.asfortranarray(np.asarray(v, dtype=np.float64).reshape(f.Lt, f.N, order="F"))
        out = np.zeros_like(v, order="F")
        lib().orc_kpm_apply_real(self._h, _p(out), _p(v))
        return out

    @property
    def active(self):
        return bool(lib().orc_kpm_active(self._h))

    @property
    def bounds(self):
        b = np.zeros(2)
        lib().orc_kpm_bounds(self._h, _p(b))
        return float(b[0]), float(b[1])

    @property
    def order(self):
        o = np.zeros(self.fdm.Lt, dtype=np.int32)
        n = lib().orc_kpm_order(self._h, _p(o))
        return o[:n].copy()

    def coefs(self, slot):
        n = int(self.order[slot])
        c = np.zeros(n, dtype=np.complex128)
        lib().orc_kpm_coefs(self._h, int(slot), _p(c))
        return c

    def lanczos(self):
        a = np.zeros(self.n)
        b = np.zeros(self.n - 1)
        lib().orc_kpm_lanczos(self._h, _p(a), _p(b))
        return a, b

    def bbar(self):
        f = self.fdm
        d, c, s = np.zeros(f.N), np.zeros(f.Nh), np.zeros(f.Nh)
        lib().orc_kpm_bbar(self._h, _p(d), _p(c), _p(s))
        return d, c, s

    def bbar_mul(self, v):
        v = np.array(v, dtype=np.complex128)
        lib().orc_kpm_bbar_mul(self._h, _p(v))
        return v


def tridiag_extremes(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    lo, hi = C.c_double(0), C.c_double(0)
    lib().orc_tridiag_extremes(_p(a), _p(b), len(a), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


# ---- force terms -----------------------------------------------------

class _ElphStruct(C.Structure):
    _fields_ = [("Nph", C.c_int), ("x", C.c_void_p), ("dtau", C.c_double), ("finite_mass", C.c_void_p),
                ("Nhol", C.c_int), ("ha", C.c_void_p), ("ha2", C.c_void_p), ("ha3", C.c_void_p), ("ha4", C.c_void_p),
                ("h_c2p", C.c_void_p), ("h_c2s", C.c_void_p), ("h_phsym", C.c_void_p),
                ("Nssh", C.c_int), ("sa", C.c_void_p), ("sa2", C.c_void_p), ("sa3", C.c_void_p), ("sa4", C.c_void_p),
                ("s_c2p", C.c_void_p), ("s_bond", C.c_void_p),
                ("sa_im", C.c_void_p), ("sa2_im", C.c_void_p), ("sa3_im", C.c_void_p), ("sa4_im", C.c_void_p)]


class OracleElph:
    """Flattened electron-phonon couplings for the force terms (what the Julia shim would build from
    ``ElectronPhononParameters``: holstein_parameters_up, ssh_parameters_up, phonon masses)."""

    def __init__(self, couplings):
        c = couplings
        f = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.x = np.asfortranarray(c.x, dtype=np.float64)
        self.keep = [self.x, f(c.finite_mass, np.int32), f(c.h_alpha, np.float64), f(c.h_alpha2, np.float64), f(c.h_alpha3, np.float64), f(c.h_alpha4, np.float64),
                     f(c.h_c2p, np.int64), f(c.h_c2s, np.int64), f(c.h_phsym, np.int32), f(c.s_alpha, np.float64), f(c.s_alpha2, np.float64), f(c.s_alpha3, np.float64),
                     f(c.s_alpha4, np.float64), np.asfortranarray(c.s_c2p, dtype=np.int64), f(c.s_bond, np.int64)]
        k = self.keep
        # T = ComplexF64: the SSH couplings are complex (ssh_parameters.α::Vector{T}); their imaginary parts ride in four more arrays
        im = [getattr(c, n, None) for n in ("s_alpha_im", "s_alpha2_im", "s_alpha3_im", "s_alpha4_im")]
        self.keep_im = [None if a is None else f(a, np.float64) for a in im] if im[0] is not None else [None] * 4
        if self.keep_im[0] is not None:
            self.keep_im = [np.zeros(len(k[9])) if a is None else a for a in self.keep_im]
        pim = [None if a is None else _p(a) for a in self.keep_im]
        self.s = _ElphStruct(self.x.shape[0], _p(k[0]), c.dtau, _p(k[1]), len(k[2]), _p(k[2]), _p(k[3]), _p(k[4]), _p(k[5]), _p(k[6]), _p(k[7]), _p(k[8]),
                             len(k[9]), _p(k[9]), _p(k[10]), _p(k[11]), _p(k[12]), _p(k[13]), _p(k[14]), *pim)


def mul_dMdx(fdm: OracleFDM, elph: OracleElph, colors, nu, u, v, out=None):
    """mul_νRe∂M∂x! (src/fermion_det_matrix_dervative.jl:2-186); returns / accumulates (Nph, Ltau)."""
    Lt, N = fdm.Lt, fdm.N
    u = fvec(u).reshape(Lt, N, order="F")
    v = fvec(v).reshape(Lt, N, order="F")
    out = np.zeros((elph.x.shape[0], Lt), order="F") if out is None else out
    cols = np.asfortranarray(colors, dtype=np.int64)
    lib().orc_mul_dMdx(fdm._h, C.byref(elph.s), _p(cols), int(cols.shape[1]), C.c_double(nu), _p(u), _p(v), _p(out))
    return out


def mul_dLdx(elph: OracleElph, Lam, nu, up, u, out=None):
    """mul_νRe∂Λ∂x! (src/holstein_shift_matrix.jl:156-201)."""
    Lam = np.asfortranarray(Lam, dtype=np.float64)
    Lt, N = Lam.shape
    up = fvec(up).reshape(Lt, N, order="F")
    u = fvec(u).reshape(Lt, N, order="F")
    out = np.zeros((elph.x.shape[0], Lt), order="F") if out is None else out
    lib().orc_mul_dLdx(C.byref(elph.s), _p(Lam), Lt, N, C.c_double(nu), _p(up), _p(u), _p(out))
    return out


def fields_from_phonons(c, V0, t0, perm):
    """``SmoQyDQMC.update!(fermion_path_integral, elph, x, +1)`` on a bare path integral — the source is
    not under /root/reference, so the functional forms are the antiderivatives of what the reference
    differentiates: ``∂V_i/∂x_p = α + 2α₂x + 3α₃x² + 4α₄x³`` (src/fermion_det_matrix_dervative.jl:281) and
    ``∂K_ji/∂Δx`` of the same form with ``K = -t``, ``Δx = x[p′] - x[p]`` (:229-233).  Pinned by the
    finite-difference test in tests/test_oracle_phonon_fields.py.  ``c`` is a ForceCouplings-like object;
    returns ``V`` (N x Ltau) and ``t`` (Nh x Ltau, FermionPathIntegral hopping order)."""
    x = np.asarray(c.x, dtype=np.float64)
    Lt = x.shape[1]
    V = np.repeat(np.asarray(V0, dtype=np.float64)[:, None], Lt, axis=1)
    # T = ComplexF64: complex bare hoppings and / or complex SSH couplings (s_alpha*_im) give a complex t
    cplx = np.iscomplexobj(t0) or getattr(c, "s_alpha_im", None) is not None
    t = np.repeat(np.asarray(t0, dtype=np.complex128 if cplx else np.float64)[:, None], Lt, axis=1)
    for k in range(len(c.h_alpha)):
        xp = x[int(c.h_c2p[k]) - 1]
        V[int(c.h_c2s[k]) - 1] += c.h_alpha[k] * xp + c.h_alpha2[k] * xp**2 + c.h_alpha3[k] * xp**3 + c.h_alpha4[k] * xp**4
    s_c2p = np.asarray(c.s_c2p)
    for k in range(len(c.s_alpha)):
        dx = x[int(s_c2p[1, k]) - 1] - x[int(s_c2p[0, k]) - 1]
        h = int(perm[int(c.s_bond[k]) - 1]) - 1  # sorted bond n is model hopping perm[n]
        t[h] -= c.s_alpha[k] * dx + c.s_alpha2[k] * dx**2 + c.s_alpha3[k] * dx**3 + c.s_alpha4[k] * dx**4
        if getattr(c, "s_alpha_im", None) is not None:
            t[h] -= 1j * (c.s_alpha_im[k] * dx + c.s_alpha2_im[k] * dx**2 + c.s_alpha3_im[k] * dx**3 + c.s_alpha4_im[k] * dx**4)
    return np.asfortranarray(V), np.asfortranarray(t)
