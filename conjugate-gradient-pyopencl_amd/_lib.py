"""ctypes binding of the HIP library (include/cgamd.h, include/clcg.h).

Fails loudly: if libcgamd.so is missing, or no HIP device is present, every
compute entry raises.  There is no CPU fallback in this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcgamd.so")
LEGACY_LIB_PATH = os.path.join(_HERE, "liboclcg.so")   # drop-in for the reference's ./build/liboclcg.so

F32, F64, C64, C128 = 0, 1, 2, 3
DTYPE_CODE = {np.dtype(np.float32): F32, np.dtype(np.float64): F64,
              np.dtype(np.complex64): C64, np.dtype(np.complex128): C128}
CODE_DTYPE = {v: k for k, v in DTYPE_CODE.items()}

MATRIX_ON_DEVICE, NO_GRAPH, UNFUSED, DIST_GRAPH, DIST_NO_OVERLAP, DIST_P2P, DIST_P2P_STAGED, DIST_SINGLE_REDUCTION, DIST_RESIDENT = 1, 2, 4, 8, 32, 64, 128, 256, 512
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_IO, ERR_COMM, ERR_STATE = range(8)      # cgamd_status (include/cgamd.h)


class CgAmdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"cgamd status {status}: {message}")
        self.status = status


_lib = None


def load():
    """Load libcgamd.so (built by `make -C conjugate-gradient-pyopencl_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64 / librccl.so.1.  Two HIP runtimes in one process do
    # not coexist ("No HIP GPUs are available" in whichever comes second), so when torch is installed it is
    # imported FIRST and libcgamd.so then binds to the copies already loaded (same SONAMEs).  Without torch
    # (plain ctypes hosts, the C CLI) the library uses /opt/rocm/lib through its RPATH.
    if os.environ.get("CGAMD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}`. This package has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, ll, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t
    pvp = ctypes.POINTER(ctypes.c_void_p)
    sig = {
        "cgamd_last_error": (ctypes.c_char_p, []),
        "cgamd_version": (ci, []),
        "cgamd_dtype_size": (sz, [ci]),
        "cgamd_tune": (ci, [ctypes.c_char_p, ci]),
        "cgamd_device_count": (ci, []),
        "cgamd_device_name": (ci, [ci, ctypes.c_char_p, sz]),
        "cgamd_ctx_create": (ci, [ci, pvp]),
        "cgamd_ctx_destroy": (ci, [vp]),
        "cgamd_ctx_set_stream": (ci, [vp, vp]),
        "cgamd_ctx_stream": (vp, [vp]),
        "cgamd_ctx_device": (ci, [vp]),
        "cgamd_ctx_synchronize": (ci, [vp]),
        "cgamd_malloc": (ci, [vp, sz, pvp]),
        "cgamd_free": (ci, [vp, vp]),
        "cgamd_memcpy_h2d": (ci, [vp, vp, vp, sz]),
        "cgamd_memcpy_d2h": (ci, [vp, vp, vp, sz]),
        "cgamd_memcpy_d2d": (ci, [vp, vp, vp, sz]),
        "cgamd_memset": (ci, [vp, vp, ci, sz]),
        "cgamd_spmv": (ci, [vp, ci, ci, ll, vp, vp, vp, vp, vp, ci]),
        "cgamd_vdot": (ci, [vp, ci, ci, vp, vp, vp, ci]),
        "cgamd_axpy": (ci, [vp, ci, ci, vp, vp, vp, ci, ci]),
        "cgamd_aypx": (ci, [vp, ci, ci, vp, vp, vp, ci]),
        "cgamd_sub": (ci, [vp, ci, ci, vp, vp, vp, ci]),
        "cgamd_solver_create": (ci, [vp, ci, ci, ll, vp, vp, vp, ci, ci, pvp]),
        "cgamd_solver_destroy": (ci, [vp]),
        "cgamd_solver_set_rhs": (ci, [vp, vp, vp, ci]),
        "cgamd_solver_set_preconditioner": (ci, [vp, vp, ci]),
        "cgamd_solver_iterate": (ci, [vp, ci]),
        "cgamd_solver_iterate_timed": (ci, [vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
        "cgamd_solver_get_x": (ci, [vp, vp, ci]),
        "cgamd_solver_history": (ci, [vp, vp, ci]),
        "cgamd_solver_iterations_done": (ci, [vp]),
        "cgamd_solver_vector": (vp, [vp, ci]),
        "cgamd_solver_ld": (ci, [vp]),
        "cgamd_solver_solve": (ci, [vp, vp, vp, ci, vp]),
        "cgamd_solver_spmv": (ci, [vp, vp, vp, ci]),
        "cgamd_solver_spmm_rowmajor": (ci, [vp, vp, vp, ci]),
        "cgamd_solver_layout": (ci, [vp]),
        "cgamd_solver_loop_launches": (ci, [vp]),
        "cgamd_solver_index_codes": (ci, [vp]),
        "cgamd_solver_value_codes": (ci, [vp]),
        "cgamd_solver_joint_codes": (ci, [vp]),
        "cgamd_solver_iterate_tol": (ci, [vp, ci, ctypes.c_double, ctypes.POINTER(ci)]),
        "cgamd_transpose": (ci, [vp, ci, ci, ci, vp, vp]),
        "cgamd_solver_spmv_bytes": (ll, [vp]),
        "cgamd_solver_iter_bytes": (ll, [vp, ci]),
        "cgamd_solver_spmv_moved_bytes": (ll, [vp]),
        "cgamd_solver_iter_moved_bytes": (ll, [vp]),
        "cgamd_cg": (ci, [ci, ci, ll, vp, vp, vp, vp, vp, ci, ci, vp, ci]),
        "cgamd_cg_last_timing": (ci, [ctypes.POINTER(ctypes.c_double)]),
        "cgamd_cg_release_cache": (ci, []),
        "cgamd_solver_reload_matrix": (ci, [vp, vp, vp, vp]),
        "cgamd_gen_laplace3d": (ci, [vp, ci, ci, ci, ci, ll, ll, vp, vp, vp, ctypes.POINTER(ll)]),
        "cgamd_gen_poisson2d": (ci, [vp, ci, ci, vp, vp, vp, ctypes.POINTER(ll)]),
        "cgamd_gen_helm_fe_var": (ci, [vp, ci, ci, ctypes.c_double, vp, ctypes.c_double, ci, ci, vp, vp, vp, ctypes.POINTER(ll)]),
        "cgamd_gen_local_rect": (ci, [vp, ci, ci, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ci, ci, vp, vp, vp,
                                 ctypes.POINTER(ll)]),
        "cgamd_gen_rhs": (ci, [vp, ci, ci, ci, ctypes.c_double, vp]),
        "cgamd_mm_read": (ci, [ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ll), ctypes.POINTER(ci),
                               ctypes.POINTER(ctypes.POINTER(ctypes.c_double)),
                               ctypes.POINTER(ctypes.POINTER(ci)), ctypes.POINTER(ctypes.POINTER(ci))]),
        "cgamd_mm_free": (None, [vp]),
        "cgamd_comm_unique_id": (ci, [vp]),
        "cgamd_dist_create": (ci, [vp, vp, ci, ci, ci, ci, ci, ll, vp, vp, vp, ci, vp, vp, vp, vp, ci, pvp]),
        "cgamd_dist_destroy": (ci, [vp]),
        "cgamd_dist_set_rhs": (ci, [vp, vp, vp]),
        "cgamd_dist_iterate": (ci, [vp, ci]),
        "cgamd_dist_get_x": (ci, [vp, vp]),
        "cgamd_dist_history": (ci, [vp, vp, ci]),
        "cgamd_dist_synchronize": (ci, [vp]),
        "cgamd_p2p_mailbox_alloc": (ci, [vp, ll, ci, pvp, vp]),
        "cgamd_p2p_mailbox_free": (ci, [vp, vp]),
        "cgamd_dist_attach_p2p": (ci, [vp, vp, vp, vp]),
        "cgamd_dist_p2p_error": (ci, [vp]),
        "cgamd_dist_enable_resident": (ci, [vp, ll, vp, vp]),
        "cgamd_dist_index_codes": (ci, [vp]),
        "cgamd_dist_loop_launches": (ci, [vp]),
        "cgamd_dist_comm_ranks": (ci, [vp]),
        # legacy entry, reference clcg.h:3-5
        "cg": (vp, [ci, ci, vp, vp, vp, vp, vp, ci, ci, ci]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise CgAmdError(status, load().cgamd_last_error().decode(errors="replace"))


def ptr(a):
    """void* of a numpy array, a torch tensor, an int address, or None."""
    if a is None:
        return None
    if isinstance(a, int):
        return ctypes.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(ctypes.c_void_p)
    if hasattr(a, "data_ptr"):          # torch.Tensor (device or host)
        return ctypes.c_void_p(a.data_ptr())
    if hasattr(a, "ptr"):               # DeviceBuffer
        return ctypes.c_void_p(a.ptr)
    raise TypeError(f"cannot take the address of {type(a)}")
