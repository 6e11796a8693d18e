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
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz"))
            for name in ("generators", "cg_iterates", "driver_generators")}


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


def rand_csr(rng, n, avg_nnz, dtype, spd=False, empty_rows=False, long_row=None):
    """Random CSR with unsorted columns inside rows (the kernels must not rely on sortedness)."""
    import scipy.sparse as sp
    density = min(1.0, avg_nnz / max(n, 1))
    A = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 31)), format="lil")
    if long_row is not None:
        r, ln = long_row
        cols = rng.choice(n, size=min(ln, n), replace=False)
        A[r, cols] = 1.0
    A = A.tocsr()
    if empty_rows:
        A = A.tolil()
        for r in rng.choice(n, size=max(1, n // 7), replace=False):
            A[r, :] = 0
        A = A.tocsr()
        A.eliminate_zeros()
    data = rand_vec(rng, A.nnz, dtype)
    A = sp.csr_matrix((data, A.indices, A.indptr), shape=(n, n))
    if spd:
        A = (A + A.T.conj() if False else A + A.T) * 0.5
        A = A + sp.identity(n, dtype=dtype) * (abs(A).sum(axis=1).max() + 1.0)
        A = sp.csr_matrix(A)
    # shuffle columns inside each row
    indptr, indices, data = A.indptr.astype(np.int32), A.indices.astype(np.int32).copy(), A.data.astype(dtype).copy()
    for r in range(n):
        s, e = indptr[r], indptr[r + 1]
        p = rng.permutation(e - s)
        indices[s:e] = indices[s:e][p]
        data[s:e] = data[s:e][p]
    return indptr, indices, data
