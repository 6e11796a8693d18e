"""Synthetic systems generated directly in HBM (bench inputs never cross PCIe).

laplace3d : 7-point 3-D Laplacian, x fastest, Dirichlet, diag 6 / off -1 -- the headline
            N=10M system (250x200x200) and the N=100M system (464^3) of SURVEY §8d.
poisson2d : 5-point 2-D Laplacian, diag 4 / off -1 == reference Poisson(N), p_h-PY_C-CL.py:1642-1682.
Both return device CSR in torch tensors (torch supplies the device memory; the kernels are ours).
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
