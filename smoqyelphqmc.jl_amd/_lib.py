"""ctypes binding of ``libsmoqy_hip.so`` (C ABI in ``include/smoqy_hip.h``).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible every entry
point raises ``SmoqyError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libsmoqy_hip.so")
MEMBER_LIB_PATH = os.path.join(CSRC, "libsmoqy_member.so")  # the member side of a published walker team alone: no HIP / rocFFT dependency
HEADER = os.path.join(os.path.dirname(_HERE), "include", "smoqy_hip.h")

OP_M, OP_MT, OP_MTM, OP_MMT = 0, 1, 2, 3
LAMBDA_MUL, LAMBDA_LDIV, LAMBDA_MULT, LAMBDA_LDIVT = 0, 1, 2, 3


class SmoqyError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the Julia shim throws the same way, so the
    reference's try/catch around force/action evaluations keeps rejecting the update,
    src/EFAPFFHMCUpdater.jl:168-187)."""


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp", ".map"))] + [HEADER, os.path.join(CSRC, "Makefile")]
    stale = force or not (os.path.exists(LIB_PATH) and os.path.exists(MEMBER_LIB_PATH)) or any(
        os.path.getmtime(s) > min(os.path.getmtime(LIB_PATH), os.path.getmtime(MEMBER_LIB_PATH)) for s in srcs)
    if stale:
        r = subprocess.run(["make", "-C", CSRC, "-j4"] + (["-B"] if force else []), capture_output=True, text=True)
        if r.returncode != 0:
            raise SmoqyError("hipcc build of libsmoqy_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_p = C.c_void_p
_i = C.c_int
_d = C.c_double
_pi = C.POINTER(C.c_int)
_pd = C.POINTER(C.c_double)

# name -> argtypes (every function returns int except smoqy_last_error)
SIGNATURES = {
    "smoqy_create": [C.POINTER(_p), _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i],
    "smoqy_destroy": [_p],
    "smoqy_clone": [C.POINTER(_p), _p, _i],
    "smoqy_set_stream": [_p, _p],
    "smoqy_sync": [_p],
    "smoqy_host_alloc": [_p, C.POINTER(_p), C.c_size_t],
    "smoqy_host_free": [_p, _p],
    "smoqy_host_register": [_p, _p, C.c_size_t],
    "smoqy_host_unregister": [_p, _p],
    "smoqy_dims": [_p, _pi],
    "smoqy_traits": [_p, _pi],
    "smoqy_describe": [_p, C.c_char_p, C.c_size_t],
    "smoqy_set_tau_chunk": [_p, _i],
    "smoqy_get_tau_chunk": [_p, _pi],
    "smoqy_update_fields": [_p, _i, _p, _p, _p],
    "smoqy_update_from_path_integral": [_p, _i, _p, _p, _p, _d],
    "smoqy_update_from_path_integral_all": [_p, _p, _p, _p, _d],
    "smoqy_get_fields": [_p, _i, _p, _p, _p],
    "smoqy_vec_alloc": [_p, _pi],
    "smoqy_vec_free": [_p, _i],
    "smoqy_vec_upload": [_p, _i, _p, _i, _i],
    "smoqy_vec_download": [_p, _i, _p, _i, _i],
    "smoqy_vec_copy": [_p, _i, _i],
    "smoqy_vec_dot": [_p, _i, _i, _p],
    "smoqy_matvec_v": [_p, _i, _i, _i],
    "smoqy_matvec_force_generic": [_p, _i],
    "smoqy_matvec_stream": [_p, _i],
    "smoqy_matvec_wave": [_p, _i],
    "smoqy_team_create": [C.POINTER(_p), _p, _i],
    "smoqy_team_destroy": [_p],
    "smoqy_team_size": [_p, _pi],
    "smoqy_team_set_timeout": [_p, _d],
    "smoqy_team_vectors": [_p, _pi, _pi],
    "smoqy_team_sample_phi": [_p, _i, _p, _pd],
    "smoqy_team_pff_step": [_p, _i, _p, _p, _d, _i, _i, _pd, _pi, _pd, _p],
    "smoqy_team_serve": [_p, C.c_char_p, _p],
    "smoqy_team_unserve": [_p],
    "smoqy_member_attach": [C.POINTER(_p), C.c_char_p, _i, _d],
    "smoqy_member_detach": [_p],
    "smoqy_member_dims": [_p, _pi],
    "smoqy_member_fields": [_p, _p],
    "smoqy_member_sample_phi": [_p, _p, _pd],
    "smoqy_member_pff_step": [_p, _p, _p, _d, _i, _i, _pd, _pi, _pd, _p],
    "smoqy_team_hmc_update": [_p, _i, _p, _p, _p, _p, _i, _d, _d, _d, _i, _p, _p, _p, _pi],
    "smoqy_team_hmc_finish": [_p, _i, _i],
    "smoqy_member_hmc_update": [_p, _p, _p, _p, _p, _i, _d, _d, _d, _i, _p, _p, _p, _pi],
    "smoqy_member_hmc_finish": [_p, _i],
    "smoqy_team_ge_config": [_p, _i, _i, _i, _p],
    "smoqy_team_ge_update": [_p, _i, _p, _p, _d, _i, _pi, _pd],
    "smoqy_team_ge_measure_GD0": [_p, _i, _i, _i, _p],
    "smoqy_member_ge_dims": [_p, _pi, C.POINTER(C.c_size_t)],
    "smoqy_member_ge_update": [_p, _p, _p, _d, _i, _pi, _pd],
    "smoqy_member_ge_measure_GD0": [_p, _i, _i, _p],
    "smoqy_team_bench_sweeps": [_p, _p, _i, _d, _i, _d, _d, _i, _i, _i, _i, C.c_ulong, _pd, C.POINTER(C.c_long), C.POINTER(C.c_long)],
    "smoqy_bench_randn": [_p, C.c_long, C.c_ulong, _d],
    "smoqy_matvec": [_p, _i, _p, _p, _i, _i],
    "smoqy_checkerboard_v": [_p, _i, _i, _i, _i, _i],
    "smoqy_checkerboard": [_p, _p, _i, _i, _i, _i, _i, _i],
    "smoqy_lambda_set": [_p, _i, _p],
    "smoqy_lambda_update": [_p, _i, _p, _i, _d, _i, _p, _p, _p, _p, _p],
    "smoqy_lambda_update_all": [_p, _p, _i, _d, _i, _p, _p, _p, _p, _p],
    "smoqy_lambda_get": [_p, _i, _p],
    "smoqy_lambda_apply_v": [_p, _i, _i, _i],
    "smoqy_lambda_apply": [_p, _i, _p, _p, _p, _i, _i],
    "smoqy_fft_use_rocfft": [_p, _i],
    "smoqy_fft_forward_v": [_p, _i],
    "smoqy_fft_inverse_v": [_p, _i],
    "smoqy_fft_forward": [_p, _p, _i, _i],
    "smoqy_fft_inverse": [_p, _p, _i, _i],
    "smoqy_precond_config": [_p, _d, _i, _d, _d],
    "smoqy_precond_update": [_p, _i, _p],
    "smoqy_precond_update_all": [_p, _p],
    "smoqy_precond_force_generic": [_p, _i],
    "smoqy_precond_get": [_p, _i, _pi, _pd, _pi, _pi, _pd, _pd],
    "smoqy_precond_get_coefs": [_p, _i, _i, _p],
    "smoqy_precond_set": [_p, _i, _i, _p, _p, _p],
    "smoqy_precond_apply_v": [_p, _i, _i],
    "smoqy_precond_apply": [_p, _p, _p, _i, _i],
    "smoqy_precond_apply_real": [_p, _p, _p, _i, _i],
    "smoqy_cg_solve_v": [_p, _i, _i, _d, _i, _i, _p, _p],
    "smoqy_cg_solve": [_p, _p, _p, _i, _i, _i, _d, _i, _i, _p, _p],
    "smoqy_cg_config": [_p, _i],
    "smoqy_cg_split": [_p, _i],
    "smoqy_tfft_form": [_p, _i],
    "smoqy_cg_gate": [_i],  # process-wide, no handle
    "smoqy_cg_use_graph": [_p, _i],
    "smoqy_cg_graph_status": [_p, _pi, _pi],
    "smoqy_force_set_couplings": [_p, _p],
    "smoqy_force_set_phonons": [_p, _p],
    "smoqy_force_dMdx_v": [_p, _d, _i, _i, _p],
    "smoqy_force_dLdx_v": [_p, _d, _i, _i, _p],
    "smoqy_force_v": [_p, _i, _p],
    "smoqy_force_store_v": [_p, _i, _p],
    "smoqy_pff_step_v": [_p, _i, _i, _p, _p, _d, _i, _i, _p, _p, _p, _p],
    "smoqy_set_bare_model": [_p, _p, _p, _p],
    "smoqy_update_from_phonons_all": [_p, _p],
    "smoqy_efa_config": [_p, _p, _p],
    "smoqy_efa_set_state": [_p, _p, _p],
    "smoqy_efa_get_state": [_p, _p, _p],
    "smoqy_efa_initialize_momentum": [_p, _p, _p],
    "smoqy_efa_energies": [_p, _p, _p],
    "smoqy_efa_evolve": [_p, _d, _d, _i],
    "smoqy_efa_checkpoint": [_p, _i],
    "smoqy_efa_restore_walkers": [_p, _p],
    "smoqy_hmc_trajectory_v": [_p, _i, _i, _i, _d, _d, _i, _i, _p, _p, _p, _p],
    "smoqy_hmc_async": [_p, _i, C.POINTER(C.c_long), C.POINTER(C.c_long)],
    "smoqy_copy_fields": [_p, _i, _p, _i],
    "smoqy_ge_config": [_p, _i, _i, _p],
    "smoqy_ge_measure_GD0": [_p, _i, _i, _i, _i, _p],
    "smoqy_ge_boundary_dot": [_p, _i, _i, _i, _i, _p, _p, _i, _p, _p, _i, _p],
    "smoqy_ge_measure_pairs": [_p, _i, _i, _p, _p, _i, _p, _i, _p],
    "smoqy_timer_start": [_p],
    "smoqy_timer_stop": [_p, _pd],
    "smoqy_matvec_timing": [_p, _i, _i],
    "smoqy_matvec_timing_read": [_p, _pd, _pi],
    "smoqy_cg_iteration_timing": [_p, _i],
    "smoqy_cg_iteration_timing_read": [_p, _p, _pi],
    "smoqy_matvec_timing_read_device": [_p, _pd, _pi],
    "smoqy_bench_copy": [_p, C.c_size_t, _i, _pd],
    "smoqy_bench_matvec": [_p, _i, _i, _i, _i, _pd],
    "smoqy_algorithmic_bytes": [_p, _i, _pd],
}

_lib = None


def load():
    """dlopen the library and attach prototypes.  Raises ``SmoqyError`` when it is missing —
    callers must not fall back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmoqyError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        if os.environ.get("SMOQY_AB_OLD_LIBRARY") and not hasattr(lib, name):
            continue  # tools/ab_libs.sh only: an older build of the library copied over the in-tree one lacks the newest entry points
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = C.c_int
    lib.smoqy_last_error.argtypes = [_p]
    lib.smoqy_last_error.restype = C.c_char_p
    lib.smoqy_team_last_error.argtypes = [_p]
    lib.smoqy_team_last_error.restype = C.c_char_p
    lib.smoqy_member_last_error.argtypes = [_p]
    lib.smoqy_member_last_error.restype = C.c_char_p
    _lib = lib
    return lib


_member_lib = None


def load_member():
    """dlopen libsmoqy_member.so — the smoqy_member_* entry points only, for a rank that joins a team another process serves and never
    touches a GPU itself (no libamdhip64 / rocFFT is mapped into such a rank).  Same prototypes as in the full library."""
    global _member_lib
    if _member_lib is not None:
        return _member_lib
    if not os.path.exists(MEMBER_LIB_PATH):
        raise SmoqyError(f"{MEMBER_LIB_PATH} is missing: run __graft_entry__.build()")
    lib = C.CDLL(MEMBER_LIB_PATH)
    for name, args in SIGNATURES.items():
        if name.startswith("smoqy_member_"):
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = C.c_int
    lib.smoqy_member_last_error.argtypes = [_p]
    lib.smoqy_member_last_error.restype = C.c_char_p
    _member_lib = lib
    return lib


def ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def as_state(a, Lt, N, count=1):
    """Check / convert ``a`` to the boundary layout: complex128, Fortran order, tau contiguous,
    ``count`` systems stacked along the last axis.  Returns an array that aliases ``a`` when
    ``a`` already has that layout (so in-place semantics work)."""
    a = np.asarray(a)
    want = Lt * N * count
    if a.size != want:
        raise ValueError(f"state vector has {a.size} elements, expected {want} (Ltau={Lt}, N={N}, count={count})")
    # f_contiguous only: a strided 1-D view (a[::2], a[::-1]) is NOT a valid boundary array — the library reads / writes Lτ·N contiguous
    # elements from the first-element pointer; contiguous 1-D arrays are f_contiguous anyway
    if a.dtype == np.complex128 and a.flags.f_contiguous:
        return a
    return np.asfortranarray(a, dtype=np.complex128)


def writable_state(a, Lt, N, count=1):
    b = as_state(a, Lt, N, count)
    if b is not a and not np.shares_memory(a, b):
        raise ValueError("output vector must be a complex128 Fortran-contiguous (tau-fastest) array")
    if not b.flags.writeable:
        raise ValueError("output vector is read-only")
    return b


class CouplingsStruct(C.Structure):
    """``smoqy_couplings`` of include/smoqy_hip.h."""

    _fields_ = [("Nph", C.c_int), ("dtau", C.c_double), ("finite_mass", C.c_void_p), ("Nholstein", C.c_int), ("h_alpha", C.c_void_p), ("h_alpha2", C.c_void_p),
                ("h_alpha3", C.c_void_p), ("h_alpha4", C.c_void_p), ("h_coupling_to_phonon", C.c_void_p), ("h_coupling_to_site", C.c_void_p), ("h_ph_sym", C.c_void_p),
                ("Nssh", C.c_int), ("s_alpha", C.c_void_p), ("s_alpha2", C.c_void_p), ("s_alpha3", C.c_void_p), ("s_alpha4", C.c_void_p), ("s_coupling_to_phonon", C.c_void_p),
                ("s_bond", C.c_void_p), ("s_alpha_im", C.c_void_p), ("s_alpha2_im", C.c_void_p), ("s_alpha3_im", C.c_void_p), ("s_alpha4_im", C.c_void_p)]


def couplings_struct(fc):
    """Build a ``smoqy_couplings`` from a ``lattice.ForceCouplings``; returns (struct, keep-alive list)."""
    f = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    k = [f(fc.finite_mass, np.int32), f(fc.h_alpha, np.float64), f(fc.h_alpha2, np.float64), f(fc.h_alpha3, np.float64), f(fc.h_alpha4, np.float64), f(fc.h_c2p, np.int64),
         f(fc.h_c2s, np.int64), f(fc.h_phsym, np.int32), f(fc.s_alpha, np.float64), f(fc.s_alpha2, np.float64), f(fc.s_alpha3, np.float64), f(fc.s_alpha4, np.float64),
         np.asfortranarray(fc.s_c2p, dtype=np.int64), f(fc.s_bond, np.int64)]
    # T = ComplexF64: imaginary parts of the SSH couplings (None / absent = real couplings)
    im = [getattr(fc, n, None) for n in ("s_alpha_im", "s_alpha2_im", "s_alpha3_im", "s_alpha4_im")]
    kim = [None] * 4 if im[0] is None else [f(np.zeros(len(k[8])) if a is None else a, np.float64) for a in im]
    s = CouplingsStruct(int(np.shape(fc.x)[0]), float(fc.dtau), ptr(k[0]), len(k[1]), ptr(k[1]), ptr(k[2]), ptr(k[3]), ptr(k[4]), ptr(k[5]), ptr(k[6]), ptr(k[7]), len(k[8]), ptr(k[8]),
                        ptr(k[9]), ptr(k[10]), ptr(k[11]), ptr(k[12]), ptr(k[13]), *[ptr(a) for a in kim])
    return s, k + kim


class GeSlot(C.Structure):
    """``smoqy_ge_slot`` of include/smoqy_hip.h."""

    _fields_ = [("source", C.c_int), ("orbital", C.c_int), ("shift", C.c_int64 * 2), ("second", C.c_int)]


class Handle:
    """Owns one ``smoqy_ctx`` (one FermionDetMatrix worth of device state per walker)."""

    def __init__(self, Lt, N, neighbor_table, color_ranges, is_sym=True, nwalkers=1, nrhs=1, device=-1, is_complex=False):
        self.lib = load()
        nt = np.asfortranarray(neighbor_table, dtype=np.int64)
        cr = np.asfortranarray(color_ranges, dtype=np.int64)
        self.Lt, self.N, self.Nh, self.ncol = int(Lt), int(N), int(nt.shape[1]) if nt.ndim == 2 else 0, int(cr.shape[1]) if cr.ndim == 2 else 0
        self.is_sym, self.nw, self.nrhs = bool(is_sym), int(nwalkers), int(nrhs)
        self.is_complex = bool(is_complex)  # matrix-element type T = ComplexF64 (complex hoppings)
        self.nsys = self.nw * self.nrhs
        self.device = int(device)
        h = _p()
        rc = self.lib.smoqy_create(C.byref(h), self.Lt, self.N, self.Nh, self.ncol, ptr(nt), ptr(cr), int(self.is_sym), int(self.is_complex), self.nw, self.nrhs, int(device))
        if rc != 0:
            raise SmoqyError(f"smoqy_create failed ({rc}): " + (self.lib.smoqy_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            for cb in getattr(self, "before_close", []):  # owners of host threads that write into this handle's page-locked buffers wait for them here
                cb()
            for p in getattr(self, "_pinned", []):
                self.lib.smoqy_host_free(self._h, p)
            self._pinned = []
            self.lib.smoqy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        rc = getattr(self.lib, name)(self._h, *args)
        if rc != 0:
            raise SmoqyError(f"{name} failed ({rc}): " + (self.lib.smoqy_last_error(self._h) or b"").decode())

    # ---- small conveniences -------------------------------------------------------------------
    def pinned_empty(self, shape, dtype=np.float64, order="C"):
        """numpy array backed by page-locked host memory owned by this handle (freed on close)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape))
        p = _p()
        self.call("smoqy_host_alloc", C.byref(p), max(n * dt.itemsize, 8))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=n).reshape(shape, order=order)

    def vec_alloc(self) -> int:
        i = C.c_int(-1)
        self.call("smoqy_vec_alloc", C.byref(i))
        return i.value

    def vec_upload(self, vid, host, sys0=0, count=None):
        count = self.nsys - sys0 if count is None else count
        a = as_state(host, self.Lt, self.N, count)
        self.call("smoqy_vec_upload", vid, ptr(a), sys0, count)

    def vec_download(self, vid, sys0=0, count=None):
        count = self.nsys - sys0 if count is None else count
        shape = (self.Lt, self.N) if count == 1 else (self.Lt, self.N, count)
        out = np.zeros(shape, dtype=np.complex128, order="F")
        self.call("smoqy_vec_download", vid, ptr(out), sys0, count)
        return out

    def vec_dot(self, a, b):
        out = np.zeros(self.nsys, dtype=np.complex128)
        self.call("smoqy_vec_dot", a, b, ptr(out))
        return out

    def traits(self):
        """smoqy_traits as a dict (what the handle's geometry selected)."""
        t = (C.c_int * 8)()
        self.call("smoqy_traits", t)
        keys = ("is_sym", "is_complex_T", "kpm_fast", "wl0", "wave_kind", "wave_lanes", "fdm_fast", "full")
        return dict(zip(keys, list(t)))

    def describe(self):
        """Kernel families of the handle's last full-batch MᵀM / Chebyshev launches and its τ-FFT form (smoqy_describe)."""
        import json

        buf = C.create_string_buffer(512)
        self.call("smoqy_describe", buf, 512)
        return json.loads(buf.value.decode())

    def algorithmic_bytes(self, op):
        v = C.c_double(0)
        self.call("smoqy_algorithmic_bytes", op, C.byref(v))
        return v.value

    def bench_matvec(self, op, out, inp, reps):
        ms = C.c_double(0)
        self.call("smoqy_bench_matvec", op, out, inp, reps, C.byref(ms))
        return ms.value

    def timer_start(self):
        self.call("smoqy_timer_start")

    def timer_stop(self):
        ms = C.c_double(0)
        self.call("smoqy_timer_stop", C.byref(ms))
        return ms.value
