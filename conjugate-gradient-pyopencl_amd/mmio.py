"""Matrix-Market ingest (reference main.c:20-33 gets this from the un-vendored BeBOP converter).

mmread() goes through the C++ reader in libcgamd.so (csrc/mmio.cpp): coordinate files,
real/integer/complex/pattern x general/symmetric/hermitian/skew-symmetric, symmetric storage
expanded, duplicates summed, 0-based canonical CSR.  mmwrite() is a small pure-Python writer
used to produce test inputs.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check


def mmread(path):
    """-> (size, indptr[int32], indices[int32], data[float64 | complex128])"""
    lib = _lib.load()
    n, nnz, cplx = ctypes.c_int(), ctypes.c_longlong(), ctypes.c_int()
    vals = ctypes.POINTER(ctypes.c_double)()
    ptr = ctypes.POINTER(ctypes.c_int)()
    cols = ctypes.POINTER(ctypes.c_int)()
    check(lib.cgamd_mm_read(str(path).encode(), ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(cplx),
                            ctypes.byref(vals), ctypes.byref(ptr), ctypes.byref(cols)))
    try:
        indptr = np.ctypeslib.as_array(ptr, shape=(n.value + 1,)).copy()
        k = max(nnz.value, 1)
        indices = np.ctypeslib.as_array(cols, shape=(k,))[:nnz.value].copy()
        raw = np.ctypeslib.as_array(vals, shape=(k * (2 if cplx.value else 1),))[:nnz.value * (2 if cplx.value else 1)].copy()
    finally:
        for p in (vals, ptr, cols):
            lib.cgamd_mm_free(ctypes.cast(p, ctypes.c_void_p))
    data = raw.view(np.complex128) if cplx.value else raw
    return n.value, indptr, indices, data


def mmwrite(path, size, rows, cols, vals, field="real", symmetry="general", comment=None):
    """Write coordinate entries (0-based rows/cols) as a Matrix-Market file."""
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {symmetry}\n")
        if comment:
            f.write(f"% {comment}\n")
        f.write(f"{size} {size} {len(rows)}\n")
        for i, j, v in zip(rows, cols, vals):
            if field == "pattern":
                f.write(f"{i + 1} {j + 1}\n")
            elif field == "complex":
                f.write(f"{i + 1} {j + 1} {np.real(v):.17g} {np.imag(v):.17g}\n")
            elif field == "integer":
                f.write(f"{i + 1} {j + 1} {int(v)}\n")
            else:
                f.write(f"{i + 1} {j + 1} {float(v):.17g}\n")
