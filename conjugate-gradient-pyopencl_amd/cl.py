"""Host-side mirror of the reference's Python solver module (reference cl.py:1-360),
driving hand-written HIP kernels through the C ABI instead of pyopencl.

Same names, argument meaning and return values as the reference for the CG path:

    initialize_cl_environment()                 reference cl.py:16-19
    initialize_cl_environment_with_device(dev)  reference cl.py:21-24
    get_gpu_devices()                           reference cl.py:26-31
    load_and_build_kernels(ctx, n_rhs)          reference cl.py:33-42
    CG(ctx, queue, kernels, size, non_zeros, a_values, b_values, a_pointers,
       a_cols, x, n_rhs, n_iterations, device=None)            reference cl.py:44-200
    conjugate_gradient_multi_gpu(<same>, device)                reference cl.py:203-360

Differences a caller can observe (all documented in INTEGRATION.md):
  * nothing is JIT-compiled: the kernels are precompiled for gfx950, so
    load_and_build_kernels is cheap and independent of n_rhs;
  * the value type follows a_values.dtype (complex64 at the reference's call sites,
    p_h-PY_C-CL.py:1926-1933; float32/float64/complex128 also work) instead of the
    module-level IS_COMPLEX switch;
  * errors raise CgAmdError instead of being printed and ignored.
There is no CPU fallback: without the HIP library or a GPU these functions raise.
"""
import ctypes
import threading

import numpy as np

from . import _lib
from ._lib import CgAmdError, check, ptr  # noqa: F401

# module constants of the reference (cl.py:5-7); values are those of the CDNA4 build
IS_COMPLEX = True
WAVE_SIZE = 64
LOCAL_SIZE = 4 * WAVE_SIZE


class Device:
    """A HIP device (stands for a pyopencl.Device)."""

    def __init__(self, index):
        self.index = int(index)
        buf = ctypes.create_string_buffer(256)
        check(_lib.load().cgamd_device_name(self.index, buf, 256))
        self.name = buf.value.decode()

    def __repr__(self):
        return f"<Device {self.index}: {self.name}>"


class Context:
    """Owns a cgamd_ctx (device + HIP stream + workspace). Stands for pyopencl.Context."""

    def __init__(self, device=0):
        self.device = device.index if isinstance(device, Device) else int(device)
        h = ctypes.c_void_p()
        check(_lib.load().cgamd_ctx_create(self.device, ctypes.byref(h)))
        self.handle = h
        self._lib = _lib.load()

    def set_stream(self, stream):
        """Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        check(self._lib.cgamd_ctx_set_stream(self.handle, ctypes.c_void_p(int(stream))))

    @property
    def stream(self):
        return self._lib.cgamd_ctx_stream(self.handle) or 0

    def synchronize(self):
        check(self._lib.cgamd_ctx_synchronize(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.cgamd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CommandQueue:
    """Stands for pyopencl.CommandQueue: the in-order HIP stream of a Context."""

    def __init__(self, ctx):
        self.ctx = ctx

    def flush(self):
        pass

    def finish(self):
        self.ctx.synchronize()


class DeviceBuffer:
    """Device memory owned through the C ABI (stands for pyopencl.Buffer)."""

    def __init__(self, ctx, nbytes=None, hostbuf=None, dtype=None):
        self.ctx = ctx
        self._lib = _lib.load()
        if hostbuf is not None:
            hostbuf = np.ascontiguousarray(hostbuf)
            nbytes = hostbuf.nbytes
            dtype = hostbuf.dtype
        self.nbytes = int(nbytes)
        self.dtype = np.dtype(dtype) if dtype is not None else np.dtype(np.uint8)
        p = ctypes.c_void_p()
        check(self._lib.cgamd_malloc(ctx.handle, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value
        if hostbuf is not None and self.nbytes:
            check(self._lib.cgamd_memcpy_h2d(ctx.handle, p, ptr(hostbuf), self.nbytes))

    def get(self, out=None):
        if out is None:
            out = np.empty(self.nbytes // self.dtype.itemsize, dtype=self.dtype)
        check(self._lib.cgamd_memcpy_d2h(self.ctx.handle, ptr(out), ctypes.c_void_p(self.ptr), self.nbytes))
        return out

    def release(self):
        if getattr(self, "ptr", None) and getattr(self.ctx, "handle", None):
            self._lib.cgamd_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def transpose(ctx, dtype, rows, cols, src, dst):
    """dst[c*rows + r] = src[r*cols + c] on the device: RHS-major [n_rhs][size] <-> row-major [size][n_rhs]"""
    check(_lib.load().cgamd_transpose(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], int(rows), int(cols), ptr(src), ptr(dst)))


def get_gpu_devices():
    """reference cl.py:26-31 -- list of GPU devices."""
    n = _lib.load().cgamd_device_count()
    if n < 0:
        raise CgAmdError(-n, _lib.load().cgamd_last_error().decode())
    return [Device(i) for i in range(n)]


def initialize_cl_environment():
    """reference cl.py:16-19 -- (ctx, queue) on the default device."""
    ctx = Context(0)
    return ctx, CommandQueue(ctx)


def initialize_cl_environment_with_device(device):
    """reference cl.py:21-24."""
    ctx = Context(device)
    return ctx, CommandQueue(ctx)


# ---- the five kernels, as callables on DeviceBuffers -------------------------------------------
_TORCH_NAMES = {"torch.float32": np.float32, "torch.float64": np.float64,
                "torch.complex64": np.complex64, "torch.complex128": np.complex128}


def _code(buf):
    dt = buf.dtype
    if str(dt) in _TORCH_NAMES:          # torch tensors are accepted wherever a DeviceBuffer is
        dt = _TORCH_NAMES[str(dt)]
    return _lib.DTYPE_CODE[np.dtype(dt)]


class _Kernels(dict):
    """{'axpy','aypx','spmv','sub','vdot'} -> callables (reference cl.py:36-42).

    spmv(queue, size, a_values, a_pointers, a_cols, x, y, n_rhs=None)
    vdot(queue, a, b, result, size, n_rhs=None)        result: DeviceBuffer of n_rhs values
    axpy(queue, x, y, a, a_sign, size, n_rhs=None)     a: DeviceBuffer of n_rhs values
    aypx(queue, x, y, a, size, n_rhs=None)
    sub(queue, a, b, result, size, n_rhs=None)
    """

    def __init__(self, ctx, n_rhs):
        lib = _lib.load()
        self.ctx, self.n_rhs = ctx, int(n_rhs)
        h = ctx.handle

        def spmv(queue, size, a_values, a_pointers, a_cols, x, y, n_rhs=None):
            nnz = (a_cols.numel() if hasattr(a_cols, "numel") else a_cols.nbytes // 4)
            check(lib.cgamd_spmv(h, _code(a_values), int(size), nnz, ptr(a_values), ptr(a_pointers), ptr(a_cols),
                                 ptr(x), ptr(y), n_rhs or self.n_rhs))

        def vdot(queue, a, b, result, size, n_rhs=None):
            check(lib.cgamd_vdot(h, _code(a), int(size), ptr(a), ptr(b), ptr(result), n_rhs or self.n_rhs))

        def axpy(queue, x, y, a, a_sign, size, n_rhs=None):
            check(lib.cgamd_axpy(h, _code(y), int(size), ptr(x), ptr(y), ptr(a), int(a_sign), n_rhs or self.n_rhs))

        def aypx(queue, x, y, a, size, n_rhs=None):
            check(lib.cgamd_aypx(h, _code(y), int(size), ptr(x), ptr(y), ptr(a), n_rhs or self.n_rhs))

        def sub(queue, a, b, result, size, n_rhs=None):
            check(lib.cgamd_sub(h, _code(a), int(size), ptr(a), ptr(b), ptr(result), n_rhs or self.n_rhs))

        super().__init__(axpy=axpy, aypx=aypx, spmv=spmv, sub=sub, vdot=vdot)


def load_and_build_kernels(ctx, n_rhs):
    """reference cl.py:33-42.  Nothing is compiled here: the gfx950 code objects ship in libcgamd.so."""
    return _Kernels(ctx, n_rhs)


# ---- persistent solver ------------------------------------------------------------------------
class Solver:
    """Matrix-resident CG handle: the reference re-uploads the matrix and re-JITs
    its kernels on every call (clcg.c:142-214, cl.py:45-46,73-84); here only b goes up and x comes down."""

    def __init__(self, ctx, size, non_zeros, a_values, a_pointers, a_cols, n_rhs=1, flags=0, dtype=None):
        self.ctx = ctx
        self._lib = _lib.load()
        on_device = bool(flags & _lib.MATRIX_ON_DEVICE)
        if not on_device:
            self.dtype = np.dtype(dtype) if dtype is not None else np.dtype(a_values.dtype)
            a_values = np.ascontiguousarray(a_values, dtype=self.dtype)
            a_pointers = np.ascontiguousarray(a_pointers, dtype=np.intc)
            a_cols = np.ascontiguousarray(a_cols, dtype=np.intc)
        else:
            if dtype is None:
                raise ValueError("dtype is required for device-resident matrices")
            self.dtype = np.dtype(dtype)
        self._keep = (a_values, a_pointers, a_cols)      # borrowed device arrays must outlive the handle
        self.size, self.n_rhs = int(size), int(n_rhs)
        h = ctypes.c_void_p()
        check(self._lib.cgamd_solver_create(ctx.handle, _lib.DTYPE_CODE[self.dtype], self.size, int(non_zeros),
                                            ptr(a_values), ptr(a_pointers), ptr(a_cols), self.n_rhs, int(flags),
                                            ctypes.byref(h)))
        self.handle = h

    def reload_matrix(self, a_values, a_pointers, a_cols):
        """new values / pattern of the same size and non-zero count from HOST arrays into a handle that owns its matrix
        (what the stateless cg() does between calls, include/cgamd.h)"""
        a_values = np.ascontiguousarray(a_values, dtype=self.dtype)
        a_pointers = np.ascontiguousarray(a_pointers, dtype=np.int32)
        a_cols = np.ascontiguousarray(a_cols, dtype=np.int32)
        check(self._lib.cgamd_solver_reload_matrix(self.handle, ptr(a_values), ptr(a_pointers), ptr(a_cols)))

    def set_rhs(self, b, x0=None, on_device=False):
        if not on_device:
            b = np.ascontiguousarray(np.asarray(b).reshape(-1), dtype=self.dtype)
            if x0 is not None:
                x0 = np.ascontiguousarray(np.asarray(x0).reshape(-1), dtype=self.dtype)
        check(self._lib.cgamd_solver_set_rhs(self.handle, ptr(b), ptr(x0), int(on_device)))

    def set_preconditioner(self, m):
        """z = m * r between residual and search direction: the reference's PCG with a diagonal `M`
        (helmFE_var.py:546-586, `z = M.dot(r)` branch).  m: the diagonal as a 1-D array (1/diag(A) for Jacobi), a scipy
        sparse diagonal matrix as the reference passes it, a device buffer, or None to go back to plain CG.  Takes effect at
        the next set_rhs; history() keeps returning r.r."""
        if m is None:
            check(self._lib.cgamd_solver_set_preconditioner(self.handle, None, 0))
            return
        if hasattr(m, "diagonal") and hasattr(m, "nnz"):          # scipy sparse: must be diagonal, as in the reference
            if m.nnz > m.shape[0]:
                raise ValueError("only a diagonal M is supported (the reference's spsolve branch is out of scope)")
            m = m.diagonal()
        on_device = not isinstance(m, np.ndarray) and not isinstance(m, (list, tuple))
        if not on_device:
            m = np.ascontiguousarray(np.asarray(m).reshape(-1), dtype=self.dtype)
            if m.size != self.size:
                raise ValueError("preconditioner diagonal must have `size` entries")
        check(self._lib.cgamd_solver_set_preconditioner(self.handle, ptr(m), int(on_device)))

    def iterate(self, n_iterations):
        check(self._lib.cgamd_solver_iterate(self.handle, int(n_iterations)))

    def iterate_timed(self, n_iterations):
        """n_iterations plain-launch iterations with HIP events around every SpMV: (avg SpMV ms, avg iteration ms)"""
        a, b = ctypes.c_float(), ctypes.c_float()
        check(self._lib.cgamd_solver_iterate_timed(self.handle, int(n_iterations), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def x(self, out=None):
        if out is None:
            out = np.empty(self.size * self.n_rhs, dtype=self.dtype)
        on_device = not isinstance(out, np.ndarray)
        check(self._lib.cgamd_solver_get_x(self.handle, ptr(out), int(on_device)))
        if on_device:
            self.ctx.synchronize()      # the copy ran on the solver's stream; make it visible to the caller's
        return out

    def iterations_done(self):
        return int(self._lib.cgamd_solver_iterations_done(self.handle))

    def history(self):
        """delta_k = r_k . r_k (unconjugated) for k = 0..iterations; shape (iterations+1, n_rhs)."""
        n = self._lib.cgamd_solver_iterations_done(self.handle) + 1
        out = np.empty((n, self.n_rhs), dtype=self.dtype)
        got = self._lib.cgamd_solver_history(self.handle, ptr(out), n)
        if got < 0:
            check(-got)
        return out[:got]

    def solve(self, b, x0=None, n_iterations=10):
        self.set_rhs(b, x0)
        self.iterate(n_iterations)
        return self.x(), self.history()

    def spmv(self, x, y, fused_dot=False):
        check(self._lib.cgamd_solver_spmv(self.handle, ptr(x), ptr(y), int(fused_dot)))

    def spmm_rowmajor(self, x, y, n_rhs):
        """Y[size][n_rhs] = A X on the matrix cores; x, y ROW-MAJOR device arrays, n_rhs in {16, 32}, f32/f64"""
        check(self._lib.cgamd_solver_spmm_rowmajor(self.handle, ptr(x), ptr(y), int(n_rhs)))

    def vector(self, which):
        """device pointer of one of the handle's own vectors; right-hand side k starts at k * self.ld values"""
        return self._lib.cgamd_solver_vector(self.handle, {"x": 0, "r": 1, "d": 2, "q": 3}[which])

    @property
    def ld(self):
        return self._lib.cgamd_solver_ld(self.handle)

    @property
    def spmv_bytes(self):
        return self._lib.cgamd_solver_spmv_bytes(self.handle)

    @property
    def index_codes(self):
        """distinct (column - row) offsets when the SpMV reads one-byte column codes instead of aCols, else 0"""
        return self._lib.cgamd_solver_index_codes(self.handle)

    @property
    def value_codes(self):
        """distinct matrix entries behind the one-byte value codes of the SpMV (0 = the kernel reads aValues)"""
        return self._lib.cgamd_solver_value_codes(self.handle)

    @property
    def joint_codes(self):
        """distinct (offset, value) pairs when the SpMV reads one joint code byte per non-zero, else 0"""
        return self._lib.cgamd_solver_joint_codes(self.handle)

    def iter_bytes(self, fused=False):
        return self._lib.cgamd_solver_iter_bytes(self.handle, int(fused))

    @property
    def spmv_moved_bytes(self):
        """bytes the handle's SpMV really moves per launch (index bytes as the kernel reads them): what a roofline fraction is priced on"""
        return self._lib.cgamd_solver_spmv_moved_bytes(self.handle)

    @property
    def iter_moved_bytes(self):
        """bytes one iteration of the handle's launched loop moves (its own vector passes, its own index bytes)"""
        return self._lib.cgamd_solver_iter_moved_bytes(self.handle)

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self._lib.cgamd_solver_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


    def solve_tol(self, b, x0=None, tol=1e-5, maxit=1000, check_every=8):
        """Tolerance-stopping variant of the reference's NumPy sub-solver (p_h-PY_C-CL.py:1338-1369, UseCG==5), which breaks
        at the first iteration with sqrt(|r.r|) < tol.  The residual history stays on the device and is read back every
        `check_every` iterations (one small read-back instead of the reference's per-iteration host reductions), more often
        as the residual nears the tolerance (_run_to_tol); when a check still overshot the first iteration that met the
        tolerance, the solve is re-run for exactly that many iterations, so x is the iterate the reference returns and not one
        from past convergence (where d.q can reach 0 and alpha 0/0).
        Returns (x, iterations_run, history).  Single right-hand side."""
        if self.n_rhs != 1:
            raise ValueError("solve_tol handles one right-hand side")
        if self._lib.cgamd_solver_loop_launches(self.handle) < 2:
            # resident loop: the stop happens on the device, in the iteration the reference stops in (no read-backs, no re-run)
            self.set_rhs(b, x0)
            run_ = ctypes.c_int(0)
            st = self._lib.cgamd_solver_iterate_tol(self.handle, int(maxit), float(tol), ctypes.byref(run_))
            if st == 0:
                return self.x(), int(run_.value), self.history()
            if st != _lib.ERR_STATE:        # only "the resident loop is not available for this call" falls back; a GPU failure is reported
                check(st)

        its = self._run_to_tol(b, x0, tol, int(maxit), int(check_every))
        return self.x(), its, self.history()

    def _run_to_tol(self, b, x0, tol, maxit, check_every):
        """Host-driven tolerance stop for the launched loops: iterate in chunks, read the residual history back after each,
        and leave x at the FIRST iteration whose sqrt(|r.r|) is below tol (or NaN).  The chunk shrinks as the residual nears
        the tolerance (the decay rate of the last chunk predicts how many iterations are left), down to single iterations,
        so the loop normally ends exactly there; when a chunk still overshot, the solve is re-run for exactly that many
        iterations.  Returns the number of iterations run."""
        self.set_rhs(b, x0)
        done, step, its = 0, max(1, check_every), maxit
        while done < maxit:
            s_ = min(step, maxit - done)
            self.iterate(s_)
            done += s_
            norms = np.sqrt(np.abs(self.history()[:, 0]))
            hit = np.nonzero(~(norms[1:] >= tol))[0]                # below tol, or NaN
            if hit.size:
                its = int(hit[0]) + 1
                break
            cur, start = float(norms[-1]), float(norms[-1 - s_])
            step = max(1, check_every)
            if 0.0 < cur < start and tol > 0:
                togo = np.log(tol / cur) / (np.log(cur / start) / s_)     # iterations left at the last chunk's rate
                if togo < 2 * step:
                    step = max(1, int(togo / 2))
        if self.iterations_done() != its:
            self.set_rhs(b, x0)
            self.iterate(its)
        return its

    def pcg(self, b, M=None, x0=None, tol=1e-6, maxit=1000, check_every=8):
        """`PCG(A, b, M, x, tol, maxit)` of the reference (helmFE_var.py:546-586) for M = None or a diagonal M: stops when
        sqrt(|r.r|) < tol, returns (x, i) with i the 0-based index of the last iteration run, like the reference.  The
        residual history stays on the device and is read back every `check_every` iterations; x is taken at the first
        iteration that met the tolerance by re-running exactly that many iterations when the check overshot it."""
        if self.n_rhs != 1:
            raise ValueError("pcg handles one right-hand side")
        self.set_preconditioner(M)
        try:
            if int(maxit) > 0 and self._lib.cgamd_solver_loop_launches(self.handle) < 2:
                self.set_rhs(b, x0)             # resident loop (with a diagonal M: the chip-wide groups): the stop happens on the device
                run_ = ctypes.c_int(0)
                st = self._lib.cgamd_solver_iterate_tol(self.handle, int(maxit), float(tol), ctypes.byref(run_))
                if st == 0:
                    return self.x(), int(run_.value) - 1
                if st != _lib.ERR_STATE:
                    check(st)
            its = self._run_to_tol(b, x0, tol, int(maxit), int(check_every))
            return self.x(), its - 1
        finally:
            self.set_preconditioner(None)


def solve_subdomains(ctx, P0, residuals, n_iterations, dtype=np.csingle, solver=None):
    """The batched sub-domain solve of the reference's Additive-Schwarz preconditioner, `as_prec` with
    UseCG in {2, 3} (p_h-PY_C-CL.py:1918-1937, 1938-1953): all n_my sub-domains share ONE matrix P[0];
    their residuals are stacked RHS-major into one b, solved with n_rhs = n_my and a fixed number of
    iterations, and split back.  P0: scipy CSR (or (indptr, indices, data)); residuals: list of arrays.
    Pass `solver` (a Solver built for P0 with n_rhs = len(residuals)) to keep the matrix resident across
    the outer GMRES iterations -- the reference re-uploads it on every call (clcg.c:202-211).
    Returns a list of complex arrays shaped like the inputs (`x[p*size:(p+1)*size].astype(complex)`)."""
    if hasattr(P0, "indptr"):
        indptr, indices, data = P0.indptr, P0.indices, P0.data
    else:
        indptr, indices, data = P0
    size = len(indptr) - 1
    n_my = len(residuals)
    b_values = np.zeros(size * n_my, dtype=dtype)
    for p in range(n_my):
        b_values[p * size:(p + 1) * size] = np.asarray(residuals[p]).ravel()
    own = solver is None
    if own:
        solver = Solver(ctx, size, len(indices), np.asarray(data, dtype=dtype), indptr, indices, n_my)
    try:
        solver.set_rhs(b_values, None)
        solver.iterate(n_iterations)
        x = solver.x()
    finally:
        if own:
            solver.close()
    return [x[p * size:(p + 1) * size].astype(complex).reshape(np.shape(residuals[p])) for p in range(n_my)]


# ---- reference entry points --------------------------------------------------------------------
def CG(ctx, queue, kernels, size, non_zeros, a_values, b_values, a_pointers, a_cols, x, n_rhs, n_iterations,
       device=None, return_history=False):
    """reference cl.py:44-200: exactly n_iterations iterations of (CO)CG on n_rhs right-hand sides.

    a_values/b_values/x: 1-D arrays of the value type (np.csingle at the reference's call sites),
    a_pointers/a_cols: np.intc.  b_values and x are RHS-major (element i of RHS r at i + r*size).
    x is the initial guess on input and receives the solution (cl.py:188); it is also returned.
    """
    dt = np.dtype(a_values.dtype)
    if dt not in _lib.DTYPE_CODE:
        raise TypeError(f"unsupported value type {dt}")
    if not (isinstance(x, np.ndarray) and x.dtype == dt and x.flags.c_contiguous):
        raise TypeError("x must be a C-contiguous numpy array of the matrix value type (it is filled in place)")
    s = Solver(ctx, size, non_zeros, a_values, a_pointers, a_cols, n_rhs)
    try:
        s.set_rhs(b_values, x)
        s.iterate(n_iterations)
        s.x(out=x.reshape(-1))
        hist = s.history() if return_history else None
    finally:
        s.close()
    return (x, hist) if return_history else x


_device_contexts = {}
_device_contexts_lock = threading.Lock()


def _context_on(device):
    """a cached Context on `device` (a Device or an index) -- used when a caller names a device that its ctx is not on"""
    index = device.index if isinstance(device, Device) else int(device)
    with _device_contexts_lock:
        ctx = _device_contexts.get(index)
        if ctx is None or not getattr(ctx, "handle", None):
            ctx = _device_contexts[index] = Context(index)
        return ctx


def conjugate_gradient_multi_gpu(ctx, queue, kernels, size, non_zeros, a_values, b_values, a_pointers, a_cols, x,
                                 n_rhs, n_iterations, device):
    """reference cl.py:203-360: the per-device worker of the RHS-sharded multi-GPU mode
    (p_h-PY_C-CL-multi-GPU.py:2123-2181).  Each device gets the whole matrix and a slice of the
    right-hand sides; there is no inter-GPU communication.  `device` selects where the solve runs: the reference passes
    the device its ctx/queue were built for (multi-GPU.py:2136-2137,2160-2163); when ctx lives on another device a cached
    context on `device` is used instead.  Thread-safe per (ctx, queue): ctypes releases the GIL during the calls, every
    handle carries its own configuration snapshot and stream, so one Python thread per device runs concurrently, as in
    the reference.  (The row-partitioned RCCL solver lives in .dist.)"""
    if device is not None:
        index = device.index if isinstance(device, Device) else int(device)
        if index != ctx.device:
            ctx = _context_on(index)
    return CG(ctx, queue, kernels, size, non_zeros, a_values, b_values, a_pointers, a_cols, x, n_rhs, n_iterations,
              device=device)


def distribute_workloads_on_devices(devices, n_subdomain):
    """reference p_h-PY_C-CL-multi-GPU.py:2123-2142: split n_subdomain right-hand sides over the devices (the first
    n_subdomain % n_gpus devices take one more) and build one (ctx, queue, kernels) per device.
    -> {device: (start, end, ctx, queue, kernels)}.  `devices` may name the same physical device several times
    (distinct Device objects): each entry gets its own context, stream and thread."""
    n_gpus = len(devices)
    tasks_per_process, extra_tasks = divmod(n_subdomain, n_gpus)
    distribution, start = {}, 0
    for i in range(n_gpus):
        end = start + tasks_per_process + (1 if i < extra_tasks else 0)
        ctx, queue = initialize_cl_environment_with_device(devices[i])
        kernels = load_and_build_kernels(ctx, end - start)
        distribution[devices[i]] = (start, end, ctx, queue, kernels)
        start = end
    return distribution


def distribute_computations_with_threads(size, non_zeros, a_values, b_values, a_pointers, a_cols, x_values, n_rhs,
                                         n_iterations, workloads):
    """reference p_h-PY_C-CL-multi-GPU.py:2144-2181: one threading.Thread per workload entry; thread `dev` solves the
    right-hand sides [start, end) with the whole matrix on its device (conjugate_gradient_multi_gpu) and copies its
    slice of the solution back into x_values under a lock.  Returns x_values.  An exception in a worker is re-raised
    here after all threads have been joined (the reference lets the thread die silently)."""
    lock = threading.Lock()
    errors = []

    def worker(dev):
        try:
            start, end, ctx, queue, kernels = workloads[dev]
            if end <= start:
                return
            b = np.ascontiguousarray(b_values[start * size:end * size])
            x = np.ascontiguousarray(x_values[start * size:end * size])
            result_x = conjugate_gradient_multi_gpu(ctx, queue, kernels, size, non_zeros, a_values, b, a_pointers, a_cols, x,
                                                    end - start, n_iterations, dev)
            with lock:
                x_values[start * size:end * size] = result_x
        except BaseException as e:      # noqa: BLE001 -- reported by the joining thread
            with lock:
                errors.append(e)

    threads = [threading.Thread(target=worker, args=(dev,)) for dev in workloads]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return x_values


def solve_rhs_sharded(devices, size, non_zeros, a_values, b_values, a_pointers, a_cols, x_values, n_rhs, n_iterations):
    """The reference's multi-GPU mode in one call (`UseCG == 6`, p_h-PY_C-CL-multi-GPU.py:1934-1944): matrix replicated,
    right-hand sides sharded over `devices`, one thread each, no inter-GPU communication."""
    workloads = distribute_workloads_on_devices(devices, n_rhs)
    try:
        return distribute_computations_with_threads(size, non_zeros, a_values, b_values, a_pointers, a_cols, x_values,
                                                    n_rhs, n_iterations, workloads)
    finally:
        for _, _, ctx, _, _ in workloads.values():
            ctx.close()


def cg(size, non_zeros, a_values, b_values, a_pointers, a_cols, x, n_rhs, n_iterations, is_complex):
    """The C entry cg() (reference clcg.h:3-5) through libcgamd.so, same argument order.
    a_values/b_values/x must be float32 (or complex64 viewed as such when is_complex)."""
    lib = _lib.load()
    lib.cg(int(size), int(non_zeros), ptr(a_values), ptr(b_values), ptr(a_pointers), ptr(a_cols), ptr(x), int(n_rhs),
           int(n_iterations), int(is_complex))
    return x
