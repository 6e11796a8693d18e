import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "conjugate-gradient-pyopencl_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    mod = importlib.import_module(PKG_NAME)
    # CG_TUNE=key=value,... runs the whole suite under a non-default kernel configuration (experiments)
    for pair in filter(None, os.environ.get("CG_TUNE", "").split(",")):
        k, v = pair.split("=")
        mod._lib.check(mod._lib.load().cgamd_tune(k.encode(), int(v)))
    return mod


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz"))
            for name in ("generators", "cg_iterates", "driver_generators", "pcg_iterates")}


@pytest.fixture(scope="session")
def gpu(pkg):
    """(ctx, queue, kernels) on device 0 -- fails loudly if the HIP library or the GPU is missing."""
    ctx, queue = pkg.initialize_cl_environment()
    kernels = pkg.load_and_build_kernels(ctx, 1)
    yield ctx, queue, kernels
    ctx.close()


ALL_DTYPES = [np.float32, np.float64, np.complex64, np.complex128]


def rand_vec(rng, n, dtype):
    dtype = np.dtype(dtype)
    v = rng.standard_normal(n)
    if dtype.kind == "c":
        v = v + 1j * rng.standard_normal(n)
    return v.astype(dtype)


def rand_csr(rng, n, avg_nnz, dtype, empty_rows=False, long_row=None):
    """Random CSR, columns unsorted inside rows (the kernels must not rely on sortedness), no duplicates."""
    counts = np.minimum(rng.poisson(avg_nnz, size=n), n)
    if empty_rows:
        counts[rng.choice(n, size=max(1, n // 7), replace=False)] = 0
    if long_row is not None:
        counts[long_row[0]] = 0
    rows = np.repeat(np.arange(n, dtype=np.int64), counts)
    cols = rng.integers(0, n, size=rows.size, dtype=np.int64)
    if long_row is not None:
        r, ln = long_row
        extra = rng.choice(n, size=min(ln, n), replace=False)
        at = int(np.searchsorted(rows, r))
        rows = np.insert(rows, at, np.full(extra.size, r))
        cols = np.insert(cols, at, extra)
    _, first = np.unique(rows * n + cols, return_index=True)
    first.sort()                                   # keep generation order => unsorted columns
    rows, cols = rows[first], cols[first]
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    return indptr, cols.astype(np.int32), rand_vec(rng, cols.size, dtype)
