"""Synthetic systems generated directly in HBM (bench inputs never cross PCIe).

laplace3d : 7-point 3-D Laplacian, x fastest, Dirichlet, diag 6 / off -1 -- the headline
            N=10M system (250x200x200) and the N=100M system (464^3) of SURVEY §8d.
poisson2d : 5-point 2-D Laplacian, diag 4 / off -1 == reference Poisson(N), p_h-PY_C-CL.py:1642-1682.
helm_fe_var : P1 finite-element Helmholtz matrix with variable wave speed == reference helmFE_var(), helmFE_var.py:9-331
            (BASELINE config 3: N = 500, omega = 12, C = 1, rho = 0.15).
local_rect  : the sub-domain matrix as_prec solves with CG == reference local_rect(), p_h-PY_C-CL.py:1439-1639.
All return device CSR in torch tensors (torch supplies the device memory; the kernels are ours).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, ptr

_TORCH_DTYPE = None


def torch_dtype(np_dtype):
    import torch
    return {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
            np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}[np.dtype(np_dtype)]


def laplace3d_nnz(nx, ny, nz, row_begin=0, row_end=None):
    lib = _lib.load()
    n = nx * ny * nz
    out = ctypes.c_longlong()
    check(lib.cgamd_gen_laplace3d(None, _lib.F64, nx, ny, nz, row_begin, n if row_end is None else row_end,
                                  None, None, None, ctypes.byref(out)))
    return out.value


def laplace3d(ctx, nx, ny, nz, dtype=np.float64, row_begin=0, row_end=None, device=None):
    """-> (indptr[int32, nloc+1], indices[int32], data) as torch CUDA tensors; GLOBAL column ids."""
    import torch
    lib = _lib.load()
    n = nx * ny * nz
    row_end = n if row_end is None else row_end
    nnz = laplace3d_nnz(nx, ny, nz, row_begin, row_end)
    dev = torch.device("cuda", ctx.device if device is None else device)
    indptr = torch.empty(row_end - row_begin + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(nnz + 16, dtype=torch.int32, device=dev)[:nnz]
    data = torch.empty(nnz + 16, dtype=torch_dtype(dtype), device=dev)[:nnz]
    torch.cuda.current_stream(dev).synchronize()
    check(lib.cgamd_gen_laplace3d(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], nx, ny, nz, row_begin, row_end,
                                  ptr(data), ptr(indptr), ptr(indices), None))
    ctx.synchronize()
    return indptr, indices, data


def poisson2d(ctx, N, dtype=np.float64, device=None):
    import torch
    lib = _lib.load()
    out = ctypes.c_longlong()
    check(lib.cgamd_gen_poisson2d(None, _lib.F64, N, None, None, None, ctypes.byref(out)))
    nnz = out.value
    dev = torch.device("cuda", ctx.device if device is None else device)
    indptr = torch.empty(N * N + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(nnz + 16, dtype=torch.int32, device=dev)[:nnz]
    data = torch.empty(nnz + 16, dtype=torch_dtype(dtype), device=dev)[:nnz]
    torch.cuda.current_stream(dev).synchronize()
    check(lib.cgamd_gen_poisson2d(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], N, ptr(data), ptr(indptr), ptr(indices), None))
    ctx.synchronize()
    return indptr, indices, data


def _fe_alloc(ctx, nnz, nn, dtype, device):
    import torch
    if np.dtype(dtype).kind != "c":
        raise ValueError("the finite-element Helmholtz matrices are complex: dtype complex64 or complex128")
    dev = torch.device("cuda", ctx.device if device is None else device)
    indptr = torch.empty(nn + 1, dtype=torch.int32, device=dev)
    indices = torch.empty(nnz + 16, dtype=torch.int32, device=dev)[:nnz]
    data = torch.empty(nnz + 16, dtype=torch_dtype(dtype), device=dev)[:nnz]
    torch.cuda.current_stream(dev).synchronize()
    return indptr, indices, data


def helm_fe_var(ctx, N, omega, C, rho, Nhoriz=None, Nvert=None, dtype=np.complex64, device=None):
    """reference helmFE_var(N, omega, C, rho, Nhoriz, Nvert): C is the (Nvert-1) x (Nhoriz-1) wave speed per mesh square
    (None = all ones) -> (indptr, indices, data) on the device, canonical CSR"""
    lib = _lib.load()
    Nhoriz = N if Nhoriz is None else Nhoriz
    Nvert = N if Nvert is None else Nvert
    Cp = None
    if C is not None:
        Carr = np.ascontiguousarray(np.asarray(C, dtype=np.float64))
        if Carr.shape != (Nvert - 1, Nhoriz - 1):
            raise ValueError(f"C must have shape (Nvert-1, Nhoriz-1) = {(Nvert - 1, Nhoriz - 1)}, got {Carr.shape}")
        Cp = ptr(Carr)
    out = ctypes.c_longlong()
    check(lib.cgamd_gen_helm_fe_var(None, _lib.C64, N, omega, None, rho, Nhoriz, Nvert, None, None, None, ctypes.byref(out)))
    indptr, indices, data = _fe_alloc(ctx, out.value, Nhoriz * Nvert, dtype, device)
    check(lib.cgamd_gen_helm_fe_var(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], N, omega, Cp, rho, Nhoriz, Nvert,
                                    ptr(data), ptr(indptr), ptr(indices), None))
    ctx.synchronize()
    return indptr, indices, data


def local_rect(ctx, N, k, eps, eta, L, Nhoriz, Nvert, dtype=np.complex64, device=None):
    """reference local_rect(N, k, eps, eta, L, Nhoriz, Nvert) -> (indptr, indices, data) on the device, canonical CSR"""
    lib = _lib.load()
    out = ctypes.c_longlong()
    check(lib.cgamd_gen_local_rect(None, _lib.C64, N, k, eps, eta, L, Nhoriz, Nvert, None, None, None, ctypes.byref(out)))
    indptr, indices, data = _fe_alloc(ctx, out.value, Nhoriz * Nvert, dtype, device)
    check(lib.cgamd_gen_local_rect(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], N, k, eps, eta, L, Nhoriz, Nvert,
                                   ptr(data), ptr(indptr), ptr(indices), None))
    ctx.synchronize()
    return indptr, indices, data


def _rhs(ctx, kind, N, k, dtype, device):
    import torch
    lib = _lib.load()
    dev = device if device is not None else torch.device("cuda", ctx.device)
    b = torch.empty(N * N, dtype=torch_dtype(dtype), device=dev)
    torch.cuda.synchronize(dev)
    check(lib.cgamd_gen_rhs(ctx.handle, _lib.DTYPE_CODE[np.dtype(dtype)], kind, N, float(k), ptr(b)))
    ctx.synchronize()
    return b


def rhs(ctx, N, k, dtype=np.complex64, device=None):
    """reference helmFE_var.rhs(N, k) (helmFE_var.py:333-368), flattened: plane-wave impedance data on the boundary nodes"""
    return _rhs(ctx, 0, N, k, dtype, device)


def rhsL(ctx, N, k, dtype=np.complex64, device=None):
    """reference helmFE_var.rhsL(N, k) (helmFE_var.py:370-377), flattened"""
    return _rhs(ctx, 1, N, k, dtype, device)


def rhsA(ctx, N, k, dtype=np.complex64, device=None):
    """reference helmFE_var.rhsA(N, k) (helmFE_var.py:379-389), flattened -- BASELINE config 3: rhsA(500, 12)"""
    return _rhs(ctx, 2, N, k, dtype, device)
