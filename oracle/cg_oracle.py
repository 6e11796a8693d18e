"""ctypes binding of oracle/_build/libcgoracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product never imports it.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libcgoracle.so")

DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1,
          np.dtype(np.complex64): 2, np.dtype(np.complex128): 3}
MODE_REFERENCE_ORDER = 0   # lane-strided + tree, WG tree + host sum
MODE_SEQUENTIAL = 3        # left-to-right sums everywhere (scipy csr_matvec order; serial dot)
MODE_FAST = 1              # sequential row sums + reference-order (work-group tree) dot: parallel, deterministic


def build(force=False):
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("cg_oracle.c", "cg_oracle_impl.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
        vp, ip, ci = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int
        _lib.cgo_spmv.argtypes = [ci, ci, vp, vp, vp, vp, vp, ci, ci]
        _lib.cgo_vdot.argtypes = [ci, ci, vp, vp, vp, ci, ci]
        _lib.cgo_axpy.argtypes = [ci, ci, vp, vp, vp, ci, ci]
        _lib.cgo_aypx.argtypes = [ci, ci, vp, vp, vp, ci]
        _lib.cgo_sub.argtypes = [ci, ci, vp, vp, vp, ci]
        _lib.cgo_cg.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, ci]
        _lib.cgo_set_threads.argtypes = [ci]
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def set_threads(n):
    return lib().cgo_set_threads(int(n))


def spmv(indptr, indices, data, x, nrhs=1, mode=MODE_REFERENCE_ORDER):
    dt = data.dtype
    n = len(indptr) - 1
    x = _c(x, dt)
    y = np.empty(n * nrhs, dtype=dt)
    rc = lib().cgo_spmv(DTYPES[dt], n, _p(_c(data, dt)), _p(_c(indptr, np.int32)),
                        _p(_c(indices, np.int32)), _p(x), _p(y), nrhs, mode)
    assert rc == 0
    return y


def vdot(a, b, nrhs=1, mode=MODE_REFERENCE_ORDER):
    dt = a.dtype
    a, b = _c(a, dt), _c(b, dt)
    n = a.size // nrhs
    out = np.empty(nrhs, dtype=dt)
    assert lib().cgo_vdot(DTYPES[dt], n, _p(a), _p(b), _p(out), nrhs, mode) == 0
    return out


def axpy(x, y, a, sign, nrhs=1):
    dt = y.dtype
    x, a = _c(x, dt), _c(np.atleast_1d(a), dt)
    y = np.array(y, dtype=dt, order="C")
    assert lib().cgo_axpy(DTYPES[dt], y.size // nrhs, _p(x), _p(y), _p(a), int(sign), nrhs) == 0
    return y


def aypx(x, y, a, nrhs=1):
    dt = y.dtype
    x, a = _c(x, dt), _c(np.atleast_1d(a), dt)
    y = np.array(y, dtype=dt, order="C")
    assert lib().cgo_aypx(DTYPES[dt], y.size // nrhs, _p(x), _p(y), _p(a), nrhs) == 0
    return y


def sub(a, b, nrhs=1):
    dt = a.dtype
    a, b = _c(a, dt), _c(b, dt)
    out = np.empty_like(a)
    assert lib().cgo_sub(DTYPES[dt], a.size // nrhs, _p(a), _p(b), _p(out), nrhs) == 0
    return out


def cg(indptr, indices, data, b, x0=None, nrhs=1, n_iterations=10, mode=MODE_REFERENCE_ORDER,
       dtype=None):
    """Reference cg() semantics (clcg.c:111-466): x in/out, exactly n_iterations.
    Returns (x, history[(n_iterations+1), nrhs])."""
    dt = np.dtype(dtype) if dtype is not None else np.dtype(data.dtype)
    n = len(indptr) - 1
    data = _c(data, dt)
    b = _c(np.asarray(b).reshape(-1), dt)
    x = np.zeros(n * nrhs, dtype=dt) if x0 is None else np.array(np.asarray(x0).reshape(-1), dtype=dt)
    hist = np.zeros((n_iterations + 1, nrhs), dtype=dt)
    rc = lib().cgo_cg(DTYPES[dt], n, int(indptr[-1]), _p(data), _p(b), _p(_c(indptr, np.int32)),
                      _p(_c(indices, np.int32)), _p(x), nrhs, n_iterations, _p(hist), mode)
    assert rc == 0
    return x, hist
