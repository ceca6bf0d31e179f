"""ctypes binding of libfsnap_hip.so (C ABI: include/fsnap_hip.h).

This is the ONLY way the Python host layer reaches the GPU.  There is no CPU fallback:
if the library is missing it is built with hipcc; if that is impossible, or no gfx950
device is present when a context is requested, a loud exception is raised.

ctypes releases the GIL for the duration of each foreign call (reference threading model:
single Python thread per rank).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int64, c_uint8, c_void_p

import numpy as np

from . import build as _build

# status codes (include/fsnap_hip.h)
OK = 0
E_ARG, E_HIP, E_STATE, E_NOMEM = -1, -2, -3, -4
NUM_NOT_SPD, NUM_SINGULAR, NUM_NONFINITE = 1, 2, 3
SOLVE_CHOL, SOLVE_LSTSQ, SOLVE_RIDGE, SOLVE_RIDGE_INV = 0, 1, 2, 3
SOLVE_LSTSQ_PROBE, SOLVE_RIDGE_PROBE, SOLVE_RIDGE_INV_PROBE = 4, 5, 6     # same, but an unresolved system comes back at once with rank = -1
PROBE_OF = {SOLVE_LSTSQ: SOLVE_LSTSQ_PROBE, SOLVE_RIDGE: SOLVE_RIDGE_PROBE, SOLVE_RIDGE_INV: SOLVE_RIDGE_INV_PROBE}
BASE_OF = {v: k for k, v in PROBE_OF.items()}
COMM_ID_BYTES = 128
REDUCE_SUM, REDUCE_MAX, REDUCE_MIN = 0, 1, 2

_P_D = POINTER(c_double)
_P_U8 = POINTER(c_uint8)

# name -> (restype, argtypes); the "exports every declared symbol" test walks this table
SIGNATURES = {
    "fsnap_version": (c_int, []),
    "fsnap_device_count": (c_int, [POINTER(c_int)]),
    "fsnap_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "fsnap_ctx_destroy": (c_int, [c_void_p]),
    "fsnap_ctx_set_stream": (c_int, [c_void_p, c_void_p]),
    "fsnap_ctx_use_own_stream": (c_int, [c_void_p]),
    "fsnap_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
    "fsnap_last_error": (c_char_p, [c_void_p]),
    "fsnap_upload_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "fsnap_bind_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "fsnap_rows_alloc": (c_int, [c_void_p, c_int64, c_int64]),
    "fsnap_drop_rows": (c_int, [c_void_p]),
    "fsnap_assemble": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int]),
    "fsnap_download_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "fsnap_set_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fsnap_set_weights_train": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "fsnap_bind_weights": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fsnap_normal_eq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fsnap_normal_eq_async": (c_int, [c_void_p, c_void_p]),
    "fsnap_normal_eq_resident": (c_int, [c_void_p, POINTER(c_void_p)]),
    "fsnap_download_packed": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "fsnap_mirror_packed": (c_int, [c_void_p, c_void_p, c_int64]),
    "fsnap_weight_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "fsnap_weight_rows_device": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "fsnap_predict": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fsnap_residual_rhs": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    "fsnap_solve": (c_int, [c_int, c_double, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_double)]),
    "fsnap_cond_info": (c_int, [c_void_p]),
    "fsnap_lasso_gram": (c_int, [c_int64, c_void_p, c_void_p, c_double, c_double, c_int64, c_double, c_void_p,
                                 POINTER(c_int64), POINTER(c_double)]),
    "fsnap_normal_eq_accumulate": (c_int, [c_void_p, c_void_p]),
    "fsnap_assemble_accumulate": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int,
                                          c_void_p]),
    "fsnap_error_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fsnap_solve_device": (c_int, [c_void_p, c_int, c_double, c_int64, c_void_p, c_void_p, POINTER(c_int), POINTER(c_double)]),
    "fsnap_fit_resident": (c_int, [c_void_p, c_int, c_double, c_void_p, POINTER(c_int), POINTER(c_double), POINTER(c_void_p)]),
    "fsnap_solve_device_rhs": (c_int, [c_void_p, c_int, c_double, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_int),
                                        POINTER(c_double)]),
    "fsnap_comm_id": (c_int, [c_void_p]),
    "fsnap_comm_id_p2p": (c_int, [c_void_p]),
    "fsnap_comm_transport": (c_int, [c_void_p, POINTER(c_int)]),
    "fsnap_comm_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "fsnap_comm_destroy": (c_int, [c_void_p]),
    "fsnap_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "fsnap_allreduce_device": (c_int, [c_void_p, c_void_p, c_int64]),
    "fsnap_allreduce_host": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "fsnap_bcast_host": (c_int, [c_void_p, c_void_p, c_int64, c_int]),
    "fsnap_allgather_host": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "fsnap_barrier": (c_int, [c_void_p]),
    "fsnap_fit_dist": (c_int, [c_void_p, c_int, c_double, c_int64, c_void_p, POINTER(c_int), POINTER(c_double), POINTER(c_void_p)]),
    "fsnap_dev_alloc": (c_int, [c_void_p, c_int64, POINTER(c_void_p)]),
    "fsnap_dev_free": (c_int, [c_void_p, c_void_p]),
    "fsnap_dev_sync": (c_int, [c_void_p]),
    "fsnap_dev_upload": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "fsnap_dev_download": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "fsnap_lstsq_rows": (c_int, [c_void_p, c_double, c_int64, c_void_p, POINTER(c_int), c_void_p]),
    "fsnap_set_dense_pinv": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fsnap_rowspace_factor": (c_int, [c_int64, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "fsnap_rowspace_solve": (c_int, [c_int64, c_void_p, c_void_p, c_double, c_void_p, POINTER(c_int), c_void_p]),
    "fsnap_timing": (c_int, [c_void_p, _P_D, c_int]),
    "fsnap_timing_history": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "fsnap_timing_count": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fsnap_timing_history_comm": (c_int, [c_void_p, c_void_p, c_int]),
    "fsnap_rowspace_chain": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_double, c_void_p, POINTER(c_int), c_void_p]),
    "fsnap_launch_info": (c_int, [c_void_p, POINTER(c_int64), c_int]),
}

_lib = None


class FsnapError(RuntimeError):
    """Runtime (HIP / state / argument) failure reported by libfsnap_hip."""


def _preload_torch_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch-ROCm wheels bundle their own libamdhip64
    (SONAME libamdhip64.so.7, requested by torch as plain ``libamdhip64.so``); if this library
    pulled in /opt/rocm's copy first, a later ``import torch`` would map a second runtime and
    its device enumeration fails ("No HIP GPUs are available").  Mapping torch's copy first
    makes both resolve to the same file, in either import order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            return
        # the RCCL that belongs to that runtime (fsnap_comm.cpp loads it lazily, on the first fsnap_comm_* call)
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("FSNAP_RCCL_PATH", rccl)


def load_library(build_if_missing: bool = True):
    """Load (building first if needed) libfsnap_hip.so and attach prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # multi-process GPU work (RCCL) shares buffers through dmabuf IPC; the legacy IPC mode is not supported by the
    # host driver of the MI355X boxes (hipIpcGetMemHandle: invalid argument).  Must be set before the HSA runtime starts
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _preload_torch_hip_runtime()
    path = _build.lib_path()
    if build_if_missing:
        path = _build.build_library()
    if not os.path.exists(path):
        raise FsnapError(f"libfsnap_hip.so not found at {path}; run `python -m fitsnap_amd.build`")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = symbol missing = broken build
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(arr):
    return None if arr is None else arr.ctypes.data_as(c_void_p)


def _f64(x, name):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError(f"{name} must be C-contiguous")
    return a


def raise_status(rc: int, msg: str):
    """Map a non-zero status to the exception class the reference would raise
    (np.linalg.LinAlgError / ValueError / MemoryError)."""
    if rc == OK:
        return
    if rc in (NUM_NOT_SPD, NUM_SINGULAR):
        raise np.linalg.LinAlgError(msg or ("matrix is not positive definite" if rc == NUM_NOT_SPD else "Singular matrix"))
    if rc == NUM_NONFINITE:
        raise ValueError(msg or "array must not contain infs or NaNs")
    if rc == E_NOMEM:
        raise MemoryError(msg or "device allocation failed")
    if rc == E_ARG:
        raise ValueError(msg or "bad argument")
    raise FsnapError(f"libfsnap_hip status {rc}: {msg}")


def device_count() -> int:
    lib = load_library()
    n = c_int(0)
    rc = lib.fsnap_device_count(byref(n))
    return n.value if rc == OK else 0


def solve(kind: int, param: float, G: np.ndarray, c: np.ndarray):
    """K x K back-solve on the host side of the library (no GPU needed).
    Returns (beta, rank, rcond_estimate)."""
    lib = load_library()
    G = _f64(G, "G")
    c = _f64(c, "c")
    K = c.shape[0]
    if G.shape != (K, K):
        raise ValueError("G must be K x K")
    beta = np.empty(K, dtype=np.float64)
    rank = c_int(0)
    rce = c_double(0.0)
    rc = lib.fsnap_solve(int(kind), float(param), K, _ptr(G), _ptr(c), _ptr(beta), byref(rank), byref(rce))
    raise_status(rc, "")
    return beta, rank.value, rce.value


def cond_info():
    """(smallest scaled pivot, lambda_min estimate from the factor, S^-1 applications, 0 host / 1 device factor) of the
    calling thread's last K x K solve (fsnap_cond_info)."""
    info = np.zeros(4)
    raise_status(load_library().fsnap_cond_info(_ptr(info)), "")
    return float(info[0]), float(info[1]), int(info[2]), int(info[3])


def lasso_gram(Q: np.ndarray, q: np.ndarray, y_norm2: float, l1_reg: float, max_iter: int = 2000, tol: float = 1.0e-4,
               start=None):
    """Cyclic coordinate descent for (1/2) w^T Q w - q^T w + l1_reg |w|_1 (fsnap_lasso_gram; host side, no GPU needed).
    Returns (w, sweeps, duality gap)."""
    lib = load_library()
    Q = _f64(Q, "Q")
    q = _f64(q, "q")
    K = q.shape[0]
    if Q.shape != (K, K):
        raise ValueError("Q must be K x K")
    w = np.zeros(K) if start is None else np.array(start, dtype=np.float64)
    if w.shape != (K,):
        raise ValueError("start must have K entries")
    nit = c_int64(0)
    gap = c_double(0.0)
    rc = lib.fsnap_lasso_gram(K, _ptr(Q), _ptr(q), float(y_norm2), float(l1_reg), int(max_iter), float(tol), _ptr(w),
                              byref(nit), byref(gap))
    raise_status(rc, "")
    return w, int(nit.value), float(gap.value)


def rowspace_chain(factors, z, rcond, active=None):
    """K x K end of the row-space solve on factors kept apart (fsnap_rowspace_chain): returns (beta, rank, info)."""
    lib = load_library()
    R = _f64(np.ascontiguousarray(np.stack([np.asarray(f, dtype=np.float64) for f in factors])), "factors")
    nfac, K = R.shape[0], R.shape[1]
    z = _f64(z, "z")
    act = None if active is None else np.ascontiguousarray(np.asarray(active, dtype=np.uint8))
    beta = np.empty(K)
    rank = c_int(0)
    info = np.zeros(4)
    raise_status(lib.fsnap_rowspace_chain(K, nfac, _ptr(R), None if act is None else _ptr(act), _ptr(z), float(rcond), _ptr(beta),
                                          byref(rank), _ptr(info)), "")
    return beta, rank.value, {"chain": info[0], "norm_bound": info[1], "inverse_norm_bound": info[2], "cond_bound": info[3]}


DENSE_PINV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int64, c_int64, POINTER(c_double), c_double, POINTER(c_double),
                                 POINTER(c_double), POINTER(c_int))


def make_dense_pinv():
    """The hook ``fsnap_set_dense_pinv`` takes (include/fsnap_hip.h): x = pinv_rcond(T) y through LAPACK's divide-and-conquer
    SVD (scipy.linalg.svd, gesdd), the decomposition cached per ``token`` -- the K x K end of dgelsd for factors too large
    for the library's Jacobi sweeps.  Returns the ctypes callback object (keep it alive while it is installed)."""
    cache = {}

    def pinv_apply(user, token, n, T_p, rcond, y_p, x_p, rank_p):
        try:
            import scipy.linalg as sl

            n = int(n)
            held = cache.get("svd")
            if held is None or held[0] != (token, n):
                from ._hostblas import blas_threads

                T = np.ctypeslib.as_array(T_p, shape=(n, n))
                with blas_threads(n):               # (a BLAS pool sized by the CPUs it SEES is throttled behind a cgroup quota)
                    U, sv, Vt = sl.svd(T, full_matrices=False, lapack_driver="gesdd")
                keep = sv > rcond * sv[0] if (n and sv[0] > 0.0) else np.zeros(n, dtype=bool)   # gelsd's cut
                held = ((token, n), U[:, keep].copy(), sv[keep].copy(), Vt[keep].copy())
                cache["svd"] = held
            _, Uk, sk, Vk = held
            y = np.ctypeslib.as_array(y_p, shape=(n,))
            x = np.ctypeslib.as_array(x_p, shape=(n,))
            x[:] = Vk.T @ ((Uk.T @ y) / sk)
            rank_p[0] = int(sk.size)
            return 0
        except Exception:       # noqa: BLE001 - any failure: the library falls back on its own SVD
            return 1

    return DENSE_PINV_FN(pinv_apply)


TRANSPORTS = ("none", "rccl", "p2p")


def comm_id(transport: str = None) -> bytes:
    """A fresh communicator id (rank 0 calls this and hands the 128 bytes to every rank).  ``transport``: "rccl", "p2p"
    (the one-shot peer-to-peer all-reduce over hipIpc windows, csrc/fsnap_p2p.cpp) or None = FSNAP_DIST_TRANSPORT, else
    RCCL.  The transport travels with the id: ``comm_init`` recognises it."""
    lib = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    if transport not in (None, "rccl", "p2p"):
        raise ValueError(f"unknown transport {transport!r}: 'rccl' or 'p2p'")
    if transport is None:
        transport = "p2p" if os.environ.get("FSNAP_DIST_TRANSPORT") == "p2p" else "rccl"
    if transport == "p2p":
        rc, name = lib.fsnap_comm_id_p2p(buf), "fsnap_comm_id_p2p"
    else:
        env = os.environ.pop("FSNAP_DIST_TRANSPORT", None)      # (fsnap_comm_id honours the variable by itself)
        try:
            rc, name = lib.fsnap_comm_id(buf), "fsnap_comm_id"
        finally:
            if env is not None:
                os.environ["FSNAP_DIST_TRANSPORT"] = env
    if rc != OK:
        raise FsnapError(f"{name}: " + (lib.fsnap_last_error(None) or b"").decode())
    return buf.raw


def rowspace_factor(G, Rhat=None, tol=1.0e-10):
    """Host step of a row-space pass (fsnap_rowspace_factor): returns (Rp or None when converged, Rhat, info) with
    info = (deviation, converged, shift).  ``Rhat=None`` starts a new factorisation."""
    lib = load_library()
    G = _f64(G, "G")
    K = G.shape[0]
    first = Rhat is None
    Rhat = np.zeros((K, K)) if first else _f64(Rhat, "Rhat")
    Rp = np.empty((K, K))
    info = np.zeros(3)
    raise_status(lib.fsnap_rowspace_factor(K, _ptr(G), int(first), float(tol), _ptr(Rhat), _ptr(Rp), _ptr(info)), "")
    return (None if info[1] else Rp), Rhat, tuple(info)


def rowspace_solve(Rhat, z, rcond=1.0e-13):
    """K x K end of the row-space solve (fsnap_rowspace_solve): returns (beta, rank, info)."""
    lib = load_library()
    Rhat = _f64(Rhat, "Rhat")
    z = _f64(z, "z")
    K = z.shape[0]
    beta = np.empty(K)
    rank = c_int(0)
    info = np.zeros(4)
    raise_status(lib.fsnap_rowspace_solve(K, _ptr(Rhat), _ptr(z), float(rcond), _ptr(beta), byref(rank), _ptr(info)), "")
    return beta, rank.value, tuple(info)


class HipContext:
    """One GPU's resident copy of (A, b, w, mask) and the kernels that run on it."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._h = c_void_p(None)
        rc = self._lib.fsnap_ctx_create(int(device), byref(self._h))
        if rc != OK:
            msg = (self._lib.fsnap_last_error(None) or b"").decode()
            raise FsnapError(f"cannot create a gfx950 context on device {device}: {msg} "
                             "(the HIP path is mandatory; there is no CPU fallback)")
        self.device = device
        self.m = 0
        self.K = 0
        self._keep = []  # keep numpy buffers alive across async copies
        self.resident_train_mask = None   # mask object last sent with set_weights_train (identity = still resident)
        # LAPACK's SVD for the large truncated K x K solves of the row-space path (the library's own is a Jacobi SVD)
        self._dense_pinv = make_dense_pinv()
        self._lib.fsnap_set_dense_pinv(self._h, ctypes.cast(self._dense_pinv, c_void_p), None)

    # -- plumbing --------------------------------------------------------------------
    def _check(self, rc):
        if rc != OK:
            raise_status(rc, (self._lib.fsnap_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.fsnap_ctx_destroy(self._h)
            self._h = c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle):
        self._check(self._lib.fsnap_ctx_set_stream(self._h, c_void_p(stream_handle or None)))

    def use_own_stream(self):
        self._check(self._lib.fsnap_ctx_use_own_stream(self._h))

    def set_option(self, key: str, value: int):
        self._check(self._lib.fsnap_set_option(self._h, key.encode(), int(value)))

    # -- rows / weights ----------------------------------------------------------------
    def upload_rows(self, A: np.ndarray, b: np.ndarray):
        A = np.asarray(A)
        if A.dtype != np.float64 or A.ndim != 2:
            A = np.ascontiguousarray(A, dtype=np.float64)
            if A.ndim != 2:
                raise ValueError("A must be 2-D")
        if A.strides[1] != 8 or A.strides[0] % 8 or A.strides[0] < A.shape[1] * 8:
            A = np.ascontiguousarray(A)
        b = _f64(b, "b")
        m, K = A.shape
        if b.shape != (m,):
            raise ValueError(f"b has shape {b.shape}, expected ({m},)")
        lda = A.strides[0] // 8
        self._check(self._lib.fsnap_upload_rows(self._h, _ptr(A), m, K, lda, _ptr(b)))
        if m != self.m:
            self.resident_train_mask = None
        self.m, self.K = m, K

    def bind_rows(self, dA_ptr: int, m: int, K: int, lda: int, db_ptr: int):
        self._check(self._lib.fsnap_bind_rows(self._h, c_void_p(dA_ptr), m, K, lda, c_void_p(db_ptr)))
        self.m, self.K = m, K

    def drop_rows(self):
        """No rows on this context any more (a rank of a multi-GPU job that owns none for the next fit)."""
        self._check(self._lib.fsnap_drop_rows(self._h))
        self.m = 0
        self.resident_train_mask = None

    def rows_alloc(self, m: int, K: int):
        self._check(self._lib.fsnap_rows_alloc(self._h, int(m), int(K)))
        self.m, self.K = int(m), int(K)

    def assemble(self, raw, row0, src_row, kind, frac, d, truth, weight, fractions, blank2J, ntypes, ncoeff, offcol):
        """Batch `_collect_lammps` transform into resident rows [row0, row0 + len(src_row))."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        src_row = np.ascontiguousarray(src_row, dtype=np.int64)
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        frac = np.ascontiguousarray(frac, dtype=np.int32)
        d = np.ascontiguousarray(d, dtype=np.float64)
        truth = np.ascontiguousarray(truth, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        fractions = np.ascontiguousarray(fractions, dtype=np.float64).reshape(-1, ntypes)
        blank2J = np.ascontiguousarray(blank2J, dtype=np.float64)
        n = len(src_row)
        if not (len(kind) == len(frac) == len(d) == len(truth) == len(weight) == n):
            raise ValueError("plan arrays must have equal length")
        self._check(self._lib.fsnap_assemble(
            self._h, _ptr(raw), raw.shape[0], raw.shape[1], n, int(row0), _ptr(src_row), _ptr(kind), _ptr(frac), _ptr(d),
            _ptr(truth), _ptr(weight), _ptr(fractions) if fractions.size else None, fractions.shape[0], _ptr(blank2J),
            int(ntypes), int(ncoeff), int(offcol)))

    def assemble_accumulate(self, raw, src_row, kind, frac, d, truth, weight, fractions, blank2J, ntypes, ncoeff, offcol,
                            d_packed_ptr: int):
        """`process_single` + `c += aw.T @ aw; d += aw.T @ bw` (transpose_trick/example.py:230-237) for a batch, with the
        rows formed in registers: the packed statistics at ``d_packed_ptr`` (device) += those of the batch."""
        raw = np.ascontiguousarray(raw, dtype=np.float64)
        src_row = np.ascontiguousarray(src_row, dtype=np.int64)
        kind = np.ascontiguousarray(kind, dtype=np.int32)
        frac = np.ascontiguousarray(frac, dtype=np.int32)
        d = np.ascontiguousarray(d, dtype=np.float64)
        truth = np.ascontiguousarray(truth, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        fractions = np.ascontiguousarray(fractions, dtype=np.float64).reshape(-1, ntypes)
        blank2J = np.ascontiguousarray(blank2J, dtype=np.float64)
        n = len(src_row)
        if not (len(kind) == len(frac) == len(d) == len(truth) == len(weight) == n):
            raise ValueError("plan arrays must have equal length")
        if raw.ndim != 2 or len(blank2J) != ntypes * (ncoeff + offcol):
            raise ValueError("raw must be 2-d and blank2J must have ntypes * (ncoeff + offcol) entries")
        self._check(self._lib.fsnap_assemble_accumulate(
            self._h, _ptr(raw), raw.shape[0], raw.shape[1], n, _ptr(src_row), _ptr(kind), _ptr(frac), _ptr(d),
            _ptr(truth), _ptr(weight), _ptr(fractions) if fractions.size else None, fractions.shape[0], _ptr(blank2J),
            int(ntypes), int(ncoeff), int(offcol), c_void_p(d_packed_ptr)))

    def download_rows(self, want_a=True, want_b=True, want_w=True, out_a=None, out_b=None, out_w=None):
        A = (out_a if out_a is not None else np.empty((self.m, self.K))) if want_a else None
        b = (out_b if out_b is not None else np.empty(self.m)) if want_b else None
        w = (out_w if out_w is not None else np.empty(self.m)) if want_w else None
        lda = (A.strides[0] // 8) if A is not None else self.K
        self._check(self._lib.fsnap_download_rows(self._h, _ptr(A), lda, _ptr(b), _ptr(w)))
        return A, b, w

    def set_weights(self, w: np.ndarray, mask=None):
        w = _f64(w, "w")
        if w.shape != (self.m,):
            raise ValueError(f"w has shape {w.shape}, expected ({self.m},)")
        mk = None
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            if mk.shape != (self.m,):
                raise ValueError(f"mask has shape {mk.shape}, expected ({self.m},)")
        self._check(self._lib.fsnap_set_weights(self._h, _ptr(w), _ptr(mk)))
        if mk is not None:
            self.resident_train_mask = None          # the device mask buffer was overwritten

    def set_weights_train(self, w_train: np.ndarray, mask=None, rank=None):
        """One weight per TRAINING row (fsnap_set_weights_train); ``mask`` (uint8, 1 = train) and ``rank`` (int32
        exclusive prefix sum of the mask) go along the first time and whenever the training set changes, ``None``
        keeps the resident ones."""
        w_train = _f64(w_train, "w")
        mk = rk = None
        if mask is not None:
            mk = np.ascontiguousarray(mask, dtype=np.uint8)
            rk = np.ascontiguousarray(rank, dtype=np.int32)
            if mk.shape != (self.m,) or rk.shape != (self.m,):
                raise ValueError(f"mask / rank must have shape ({self.m},)")
        self._check(self._lib.fsnap_set_weights_train(self._h, _ptr(w_train), w_train.shape[0], _ptr(mk), _ptr(rk)))
        if mask is not None:
            self.resident_train_mask = mask          # the object whose content is on the device now

    def bind_weights(self, dw_ptr: int, dmask_ptr: int = 0):
        self._check(self._lib.fsnap_bind_weights(self._h, c_void_p(dw_ptr), c_void_p(dmask_ptr or None)))

    # -- hot path ----------------------------------------------------------------------
    def normal_eq(self):
        """Returns (G, c, scalars) on the host; scalars = [bw.bw, sum(bw), n_train]."""
        K = self.K
        G = np.empty((K, K))
        c = np.empty(K)
        s = np.empty(3)
        self._check(self._lib.fsnap_normal_eq(self._h, _ptr(G), _ptr(c), _ptr(s)))
        return G, c, s

    def normal_eq_async(self, d_packed_ptr: int):
        self._check(self._lib.fsnap_normal_eq_async(self._h, c_void_p(d_packed_ptr)))

    def normal_eq_resident(self) -> int:
        """Statistics into the context-owned device buffer; returns its device address."""
        ptr = c_void_p(None)
        self._check(self._lib.fsnap_normal_eq_resident(self._h, byref(ptr)))
        return ptr.value

    def mirror_packed(self, d_packed_ptr: int, K: int):
        """Statistics in HBM (e.g. just all-reduced) -> page-locked host mirror, asynchronously; the next
        solve_device on the same pointer then needs no D2H copy."""
        self._check(self._lib.fsnap_mirror_packed(self._h, c_void_p(d_packed_ptr), int(K)))

    def download_packed(self, d_packed_ptr: int, K: int):
        G = np.empty((K, K))
        c = np.empty(K)
        s = np.empty(3)
        self._check(self._lib.fsnap_download_packed(self._h, c_void_p(d_packed_ptr), K, _ptr(G), _ptr(c), _ptr(s)))
        return G, c, s

    def weight_rows(self):
        aw = np.empty((self.m, self.K))
        bw = np.empty(self.m)
        self._check(self._lib.fsnap_weight_rows(self._h, _ptr(aw), self.K, _ptr(bw)))
        return aw, bw

    def weight_rows_device(self, d_aw_ptr: int, ldaw: int, d_bw_ptr: int):
        self._check(self._lib.fsnap_weight_rows_device(self._h, c_void_p(d_aw_ptr), ldaw, c_void_p(d_bw_ptr)))

    def predict(self, beta, want_preds=True, want_sse=False):
        beta = _f64(beta, "beta")
        if beta.shape != (self.K,):
            raise ValueError(f"beta has shape {beta.shape}, expected ({self.K},)")
        preds = np.empty(self.m) if want_preds else None
        sse = c_double(0.0)
        self._check(self._lib.fsnap_predict(self._h, _ptr(beta), _ptr(preds), byref(sse) if want_sse else None))
        return preds, (sse.value if want_sse else None)

    def residual_rhs(self, beta, want_sse=False):
        """s = (wA)^T (wb - wA beta) on the resident rows (refinement right-hand side)."""
        beta = _f64(beta, "beta")
        if beta.shape != (self.K,):
            raise ValueError(f"beta has shape {beta.shape}, expected ({self.K},)")
        s = np.empty(self.K)
        sse = c_double(0.0)
        self._check(self._lib.fsnap_residual_rhs(self._h, _ptr(beta), _ptr(s), byref(sse) if want_sse else None))
        return s, (sse.value if want_sse else None)

    def normal_eq_accumulate(self, d_packed_ptr: int):
        """d_packed (device) += statistics of the resident rows (streaming / transpose-trick accumulation)."""
        self._check(self._lib.fsnap_normal_eq_accumulate(self._h, c_void_p(d_packed_ptr)))

    def error_stats(self, beta, cat, ncat: int):
        """Per-category sums of ``Solver.error_analysis`` for the resident rows (see fsnap_error_stats):
        returns an (ncat, 10) array."""
        beta = _f64(beta, "beta").reshape(-1)
        if beta.shape[0] != self.K:
            raise ValueError("beta must have K entries")
        if cat is not None:
            cat = np.ascontiguousarray(cat, dtype=np.int32)
            if cat.shape != (self.m,):
                raise ValueError("cat must have one entry per row")
            self.cat_serial = getattr(self, "cat_serial", 0) + 1    # identifies the categories now on the device
        stats = np.empty((int(ncat), 10), dtype=np.float64)
        self._check(self._lib.fsnap_error_stats(self._h, _ptr(beta), _ptr(cat), int(ncat), _ptr(stats)))
        return stats

    def fit_resident(self, kind: int, param: float):
        """Statistics + K x K solve of the resident rows in one library call; returns
        (beta, rank, rcond_estimate, device address of the packed statistics)."""
        beta = np.empty(self.K, dtype=np.float64)
        rank = c_int(0)
        rce = c_double(0.0)
        ptr = c_void_p()
        rc = self._lib.fsnap_fit_resident(self._h, int(kind), float(param), _ptr(beta), byref(rank), byref(rce), byref(ptr))
        if rc != OK:
            raise_status(rc, (self._lib.fsnap_last_error(self._h) or b"").decode() if rc < 0 else "")
        return beta, rank.value, rce.value, ptr.value

    def solve_device(self, kind: int, param: float, K: int, d_packed_ptr: int, rhs=None):
        """K x K solve from the packed statistics in HBM; returns (beta, rank, rcond_estimate).
        ``rhs`` (host, K doubles) replaces the c part of the packed buffer as right-hand side."""
        beta = np.empty(K, dtype=np.float64)
        rank = c_int(0)
        rce = c_double(0.0)
        if rhs is None:
            rc = self._lib.fsnap_solve_device(self._h, int(kind), float(param), int(K), c_void_p(d_packed_ptr), _ptr(beta),
                                              byref(rank), byref(rce))
        else:
            r = _f64(rhs, "rhs")
            if r.shape != (K,):
                raise ValueError("rhs must have K entries")
            rc = self._lib.fsnap_solve_device_rhs(self._h, int(kind), float(param), int(K), c_void_p(d_packed_ptr), _ptr(r),
                                                  _ptr(beta), byref(rank), byref(rce))
        if rc != OK:
            raise_status(rc, (self._lib.fsnap_last_error(self._h) or b"").decode() if rc < 0 else "")
        return beta, rank.value, rce.value

    # -- row-space least squares --------------------------------------------------------
    def lstsq_rows(self, rcond: float, K: int = None):
        """``lstsq(aw, bw, rcond)`` of the resident rows computed on the rows (fsnap_lstsq_rows); collective when the
        context has a communicator.  Returns (beta, rank, info dict)."""
        K = self.K if K is None else int(K)
        beta = np.empty(K, dtype=np.float64)
        rank = c_int(0)
        info = np.zeros(8)
        rc = self._lib.fsnap_lstsq_rows(self._h, float(rcond), K, _ptr(beta), byref(rank), _ptr(info))
        if rc != OK:
            raise_status(rc, (self._lib.fsnap_last_error(self._h) or b"").decode())
        keys = ("passes", "deviation", "converged", "svd", "sigma_max", "sigma_min", "refine_step", "shift")
        return beta, rank.value, dict(zip(keys, info.tolist()))

    # -- multi-GPU (native RCCL) ----------------------------------------------------------
    def comm_init(self, nranks: int, rank: int, ident: bytes):
        if len(ident) != COMM_ID_BYTES:
            raise ValueError("communicator id must be 128 bytes")
        self._check(self._lib.fsnap_comm_init(self._h, int(nranks), int(rank), ctypes.c_char_p(ident)))

    def comm_destroy(self):
        self._check(self._lib.fsnap_comm_destroy(self._h))

    def comm_transport(self) -> str:
        """"none" | "rccl" | "p2p": what this context's communicator runs on."""
        t = c_int(0)
        self._check(self._lib.fsnap_comm_transport(self._h, byref(t)))
        return TRANSPORTS[t.value]

    def comm_info(self):
        n, r = c_int(1), c_int(0)
        self._check(self._lib.fsnap_comm_info(self._h, byref(n), byref(r)))
        return n.value, r.value

    def allreduce_device(self, d_ptr: int, n: int):
        self._check(self._lib.fsnap_allreduce_device(self._h, c_void_p(d_ptr), int(n)))

    def allreduce_host(self, arr: np.ndarray, op: int = REDUCE_SUM):
        """In-place reduction of a C-contiguous float64 array over the ranks."""
        if arr.dtype != np.float64 or not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("allreduce_host needs a C-contiguous float64 array")
        if arr.size:
            self._check(self._lib.fsnap_allreduce_host(self._h, _ptr(arr), arr.size, int(op)))
        return arr

    def bcast_bytes(self, data: bytes, nbytes: int, root: int = 0) -> bytes:
        buf = ctypes.create_string_buffer(data if data is not None else b"", int(nbytes))
        self._check(self._lib.fsnap_bcast_host(self._h, buf, int(nbytes), int(root)))
        return buf.raw

    def allgather_bytes(self, data: bytes, nranks: int) -> list:
        """Equal-size byte strings of all ranks, in rank order."""
        n = len(data)
        out = ctypes.create_string_buffer(n * nranks)
        self._check(self._lib.fsnap_allgather_host(self._h, ctypes.c_char_p(data), n, out))
        raw = out.raw
        return [raw[i * n:(i + 1) * n] for i in range(nranks)]

    def barrier(self):
        self._check(self._lib.fsnap_barrier(self._h))

    def fit_dist(self, kind: int, param: float, K: int):
        """One multi-GPU fit (fsnap_fit_dist): local statistics, in-place RCCL all-reduce, solve on every rank.
        Returns (beta, rank, rcond_estimate, device address of the reduced statistics)."""
        beta = np.empty(int(K), dtype=np.float64)
        rank = c_int(0)
        rce = c_double(0.0)
        ptr = c_void_p()
        rc = self._lib.fsnap_fit_dist(self._h, int(kind), float(param), int(K), _ptr(beta), byref(rank), byref(rce), byref(ptr))
        if rc != OK:
            raise_status(rc, (self._lib.fsnap_last_error(self._h) or b"").decode() if rc < 0 else "")
        return beta, rank.value, rce.value, ptr.value

    # -- raw device memory ----------------------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        ptr = c_void_p()
        self._check(self._lib.fsnap_dev_alloc(self._h, int(nbytes), byref(ptr)))
        return ptr.value

    def dev_free(self, d_ptr: int):
        self._check(self._lib.fsnap_dev_free(self._h, c_void_p(d_ptr)))

    def dev_upload(self, d_ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(self._lib.fsnap_dev_upload(self._h, c_void_p(d_ptr), _ptr(arr), arr.nbytes))

    def dev_download(self, d_ptr: int, out: np.ndarray):
        if not out.flags["C_CONTIGUOUS"]:
            raise ValueError("dev_download needs a C-contiguous array")
        self._check(self._lib.fsnap_dev_download(self._h, _ptr(out), c_void_p(d_ptr), out.nbytes))
        return out

    def sync(self):
        self._check(self._lib.fsnap_dev_sync(self._h))

    # -- measurement -------------------------------------------------------------------
    def timing(self, n: int = 8):
        """HIP-event timings (ms); ``n`` = how many leading entries to evaluate (2 = kernels of the last fit only)."""
        ms = (c_double * 8)()
        self._check(self._lib.fsnap_timing(self._h, ms, int(n)))
        return {"syrk_ms": ms[0], "reduce_ms": ms[1], "upload_ms": ms[2], "weight_ms": ms[3], "predict_ms": ms[4],
                "upload_probe_GBps": ms[5], "upload_staged": bool(ms[6])}

    def timing_history(self, n: int):
        """(syrk_ms, reduce_ms) arrays of the last ``n`` fits (oldest first), read from HIP events after the fact."""
        a = np.empty(int(n))
        b = np.empty(int(n))
        self._check(self._lib.fsnap_timing_history(self._h, _ptr(a), _ptr(b), int(n)))
        return a, b

    def timing_history_comm(self, n: int):
        """Collective time (ms) on this rank's stream for the last ``n`` event-bracketed fits, -1 where a fit had none."""
        a = np.empty(int(n))
        self._check(self._lib.fsnap_timing_history_comm(self._h, _ptr(a), int(n)))
        return a

    def timing_count(self):
        """(event-bracketed SYRK launches, all SYRK launches) of this context so far (option ``timing_every``)."""
        a, b = c_int64(0), c_int64(0)
        self._check(self._lib.fsnap_timing_count(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def launch_info(self):
        info = (c_int64 * 8)()
        self._check(self._lib.fsnap_launch_info(self._h, info, 8))
        out = {"workgroups": info[0], "threads": info[1], "chunks_per_wave": info[2], "NB": info[3],
               "split": info[4], "compute_units": info[5], "kernel_or_pairs": info[6], "nsplit": info[7]}
        if info[4] != 0:            # not the tiled kernel: the last slot says whether kernel 1A packs its rows' weights itself
            out["nsplit"] = 0
            out["fused_pack"] = info[7]
        return out
