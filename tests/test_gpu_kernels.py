"""GPU parity of the five hot-path kernels against the CPU oracle, through the C ABI.

Tolerances (stated per SURVEY §8c): the HIP kernels sum in a different order than the reference
(sequential per row / wave-shuffle trees) and contract a*b+c into FMAs, so results agree to
rounding: rtol 1e-5 for f32/c64 element-wise ops and row sums, 1e-13 for f64/c128; dots
accumulate in fp64, so f32 dots are compared at 1e-6 relative to sum|a||b|.
"""
import numpy as np
import pytest

import cg_oracle
from conftest import ALL_DTYPES, rand_csr, rand_vec

pytestmark = pytest.mark.gpu

RTOL = {np.dtype(np.float32): 2e-5, np.dtype(np.complex64): 2e-5,
        np.dtype(np.float64): 1e-13, np.dtype(np.complex128): 1e-13}


def _buf(pkg, ctx, a):
    return pkg.DeviceBuffer(ctx, hostbuf=a)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("n,avg,nrhs", [(1, 1, 1), (7, 3, 1), (255, 5, 1), (256, 7, 1), (257, 7, 2), (1000, 7, 3),
                                         (5000, 9, 1), (4099, 40, 2), (70000, 7, 1)])
def test_spmv_matches_oracle(pkg, gpu, dtype, n, avg, nrhs):
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n * 31 + nrhs)
    indptr, indices, data = rand_csr(rng, n, avg, dtype, empty_rows=n > 100)
    x = rand_vec(rng, n * nrhs, dtype)
    want = cg_oracle.spmv(indptr, indices, data, x, nrhs=nrhs, mode=cg_oracle.MODE_SEQUENTIAL)
    y = _buf(pkg, ctx, np.full(n * nrhs, 7, dtype=dtype))
    kernels["spmv"](queue, n, _buf(pkg, ctx, data), _buf(pkg, ctx, indptr), _buf(pkg, ctx, indices),
                    _buf(pkg, ctx, x), y, n_rhs=nrhs)
    got = y.get()
    import scipy.sparse as sp
    scale = np.abs(sp.csr_matrix((np.abs(data), indices, indptr), shape=(n, n))) @ np.abs(x.reshape(nrhs, n).T)
    scale = scale.T.reshape(-1) + 1e-30
    assert np.max(np.abs(got - want) / scale) < RTOL[np.dtype(dtype)]


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
def test_spmv_long_rows_span_chunks(pkg, gpu, dtype):
    """rows longer than one 2048-entry LDS chunk, next to empty rows (multi-chunk accumulate path)"""
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(5)
    n = 6000
    indptr, indices, data = rand_csr(rng, n, 3, dtype, empty_rows=True, long_row=(300, 5000))
    x = rand_vec(rng, n, dtype)
    want = cg_oracle.spmv(indptr, indices, data, x, mode=cg_oracle.MODE_SEQUENTIAL)
    y = _buf(pkg, ctx, np.zeros(n, dtype=dtype))
    kernels["spmv"](queue, n, _buf(pkg, ctx, data), _buf(pkg, ctx, indptr), _buf(pkg, ctx, indices), _buf(pkg, ctx, x), y)
    got = y.get()
    assert np.max(np.abs(got - want)) / np.max(np.abs(want)) < (1e-4 if np.dtype(dtype).itemsize <= 8 and np.dtype(dtype).kind == 'c' else 1e-12)


def test_spmv_reference_order_oracle_agrees(pkg, gpu):
    """the reference's lane-strided + tree order (spmv.cl:22-43) and the sequential order agree to rounding"""
    rng = np.random.default_rng(9)
    indptr, indices, data = rand_csr(rng, 500, 40, np.float64)
    x = rand_vec(rng, 500, np.float64)
    a = cg_oracle.spmv(indptr, indices, data, x, mode=cg_oracle.MODE_REFERENCE_ORDER)
    b = cg_oracle.spmv(indptr, indices, data, x, mode=cg_oracle.MODE_SEQUENTIAL)
    assert np.allclose(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("n,nrhs", [(1, 1), (3, 2), (255, 1), (256, 3), (1001, 2), (100003, 1), (262144, 2)])
def test_vdot_unconjugated(pkg, gpu, dtype, n, nrhs):
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n + nrhs)
    a, b = rand_vec(rng, n * nrhs, dtype), rand_vec(rng, n * nrhs, dtype)
    res = _buf(pkg, ctx, np.zeros(nrhs, dtype=dtype))
    kernels["vdot"](queue, _buf(pkg, ctx, a), _buf(pkg, ctx, b), res, n, n_rhs=nrhs)
    got = res.get()
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    want = np.array([np.dot(a[r * n:(r + 1) * n].astype(wide), b[r * n:(r + 1) * n].astype(wide)) for r in range(nrhs)])
    scale = np.array([np.dot(np.abs(a[r * n:(r + 1) * n]).astype(np.float64), np.abs(b[r * n:(r + 1) * n]).astype(np.float64)) for r in range(nrhs)])
    tol = 2e-7 if np.dtype(dtype).itemsize <= 8 and np.dtype(dtype) != np.float64 else 1e-14
    assert np.max(np.abs(got - want) / scale) < tol
    # and against the C oracle in the reference's own summation order, at the reference's precision
    ref = cg_oracle.vdot(a, b, nrhs=nrhs, mode=cg_oracle.MODE_REFERENCE_ORDER)
    assert np.max(np.abs(got - ref) / scale) < (5e-6 if np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64)) else 1e-13)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("n,nrhs", [(1, 1), (5, 3), (257, 2), (4096, 1), (100001, 2)])
def test_axpy_aypx_sub(pkg, gpu, dtype, n, nrhs):
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n * 7 + nrhs)
    x, y = rand_vec(rng, n * nrhs, dtype), rand_vec(rng, n * nrhs, dtype)
    a = rand_vec(rng, nrhs, dtype)
    tol = RTOL[np.dtype(dtype)]
    for sign in (1, 0):
        yb = _buf(pkg, ctx, y)
        kernels["axpy"](queue, _buf(pkg, ctx, x), yb, _buf(pkg, ctx, a), sign, n, n_rhs=nrhs)
        want = cg_oracle.axpy(x, y, a, sign, nrhs=nrhs)
        assert np.allclose(yb.get(), want, rtol=tol, atol=tol * 4)
    yb = _buf(pkg, ctx, y)
    kernels["aypx"](queue, _buf(pkg, ctx, x), yb, _buf(pkg, ctx, a), n, n_rhs=nrhs)
    assert np.allclose(yb.get(), cg_oracle.aypx(x, y, a, nrhs=nrhs), rtol=tol, atol=tol * 4)
    rb = _buf(pkg, ctx, np.zeros_like(x))
    kernels["sub"](queue, _buf(pkg, ctx, x), _buf(pkg, ctx, y), rb, n, n_rhs=nrhs)
    assert np.array_equal(rb.get(), cg_oracle.sub(x, y, nrhs=nrhs))    # a single rounding: bit-exact


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,avg,nrhs", [(16, 3, 16), (100, 5, 32), (257, 7, 16), (1000, 7, 32), (5003, 12, 32), (40000, 5, 16)])
def test_spmm_mfma_rowmajor_matches_oracle(pkg, gpu, dtype, n, avg, nrhs):
    """config 4: SpMM with a row-major RHS block on the matrix cores (v_mfma_*_16x16x4), against the oracle's
    per-RHS SpMV; also checks the RHS-major <-> row-major transposes around it"""
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n + nrhs)
    indptr, indices, data = rand_csr(rng, n, avg, dtype, empty_rows=n > 50)
    X = rand_vec(rng, n * nrhs, dtype)                                  # RHS-major [nrhs][n]
    want = cg_oracle.spmv(indptr, indices, data, X, nrhs=nrhs, mode=cg_oracle.MODE_SEQUENTIAL)
    s = pkg.Solver(ctx, n, len(indices), data, indptr, indices, 1)
    xb, xt = _buf(pkg, ctx, X), _buf(pkg, ctx, np.zeros_like(X))
    yt, yb = _buf(pkg, ctx, np.zeros_like(X)), _buf(pkg, ctx, np.zeros_like(X))
    pkg.cl.transpose(ctx, dtype, nrhs, n, xb, xt)                       # -> [n][nrhs]
    assert np.array_equal(xt.get().reshape(n, nrhs), X.reshape(nrhs, n).T)
    s.spmm_rowmajor(xt, yt, nrhs)
    pkg.cl.transpose(ctx, dtype, n, nrhs, yt, yb)                       # -> [nrhs][n]
    got = yb.get()
    s.close()
    import scipy.sparse as sp
    scale = (np.abs(sp.csr_matrix((np.abs(data), indices, indptr), shape=(n, n))) @ np.abs(X.reshape(nrhs, n).T)).T.reshape(-1) + 1e-30
    assert np.max(np.abs(got - want) / scale) < RTOL[np.dtype(dtype)]


def test_unaligned_pointers_take_the_scalar_paths(pkg, gpu):
    """device pointers that are only 8-byte aligned (e.g. views into a larger allocation) must still be correct:
    the 16-byte-load kernels are replaced by their scalar forms (VEC = false / generic SpMV)"""
    import ctypes
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    rng = np.random.default_rng(8)
    n, nrhs = 1003, 2
    dtype = np.float64
    indptr, indices, data = rand_csr(rng, n, 6, dtype, empty_rows=True)
    x, y = rand_vec(rng, n * nrhs, dtype), rand_vec(rng, n * nrhs, dtype)
    a = rand_vec(rng, nrhs, dtype)

    def shifted(arr, off_bytes):
        big = pkg.DeviceBuffer(ctx, nbytes=arr.nbytes + 64, dtype=arr.dtype)
        pkg._lib.check(lib.cgamd_memcpy_h2d(ctx.handle, ctypes.c_void_p(big.ptr + off_bytes), pkg._lib.ptr(arr), arr.nbytes))
        return big, big.ptr + off_bytes

    keep = []
    def dev(arr, off):
        b, p = shifted(np.ascontiguousarray(arr), off)
        keep.append(b)
        return p
    pv, pc, pp = dev(data, 8), dev(indices, 4), dev(indptr, 0)
    px, py, pa = dev(x, 8), dev(y, 8), dev(a, 0)
    pkg._lib.check(lib.cgamd_spmv(ctx.handle, 1, n, len(indices), ctypes.c_void_p(pv), ctypes.c_void_p(pp), ctypes.c_void_p(pc),
                                  ctypes.c_void_p(px), ctypes.c_void_p(py), nrhs))
    got = np.empty(n * nrhs, dtype=dtype)
    pkg._lib.check(lib.cgamd_memcpy_d2h(ctx.handle, pkg._lib.ptr(got), ctypes.c_void_p(py), got.nbytes))
    want = cg_oracle.spmv(indptr, indices, data, x, nrhs=nrhs, mode=cg_oracle.MODE_SEQUENTIAL)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    # axpy / vdot on misaligned vectors with an odd leading dimension
    py2 = dev(y, 8)
    pkg._lib.check(lib.cgamd_axpy(ctx.handle, 1, n, ctypes.c_void_p(px), ctypes.c_void_p(py2), ctypes.c_void_p(pa), 1, nrhs))
    pkg._lib.check(lib.cgamd_memcpy_d2h(ctx.handle, pkg._lib.ptr(got), ctypes.c_void_p(py2), got.nbytes))
    assert np.allclose(got, cg_oracle.axpy(x, y, a, 1, nrhs=nrhs), rtol=1e-13)
    res = pkg.DeviceBuffer(ctx, hostbuf=np.zeros(nrhs, dtype=dtype))
    pkg._lib.check(lib.cgamd_vdot(ctx.handle, 1, n, ctypes.c_void_p(px), ctypes.c_void_p(py2), ctypes.c_void_p(res.ptr), nrhs))
    wd = [np.dot(x[r * n:(r + 1) * n], got[r * n:(r + 1) * n]) for r in range(nrhs)]
    assert np.allclose(res.get(), wd, rtol=1e-12)
    # a solver on a borrowed, misaligned device matrix falls back to the generic kernel and stays correct
    s = pkg.Solver(ctx, n, len(indices), pv, pp, pc, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
    b = rand_vec(rng, n, dtype)
    A_ok = None
    xs, h = s.solve(b, None, 5)
    s.close()
    xo, ho = cg_oracle.cg(indptr, indices, data, b, n_iterations=5, mode=cg_oracle.MODE_SEQUENTIAL)
    # (random non-symmetric matrix: the recurrence diverges and amplifies rounding, compare the first steps)
    assert np.allclose(h[:4, 0], ho[:4, 0], rtol=1e-9)


def test_error_reporting_on_gpu(pkg, gpu):
    ctx, queue, kernels = gpu
    ip = np.array([0, 2, 1], dtype=np.int32)        # not monotone
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.Solver(ctx, 2, 1, np.ones(1), ip, np.zeros(1, dtype=np.int32), 1)
    assert e.value.status == 1 and "monotone" in str(e.value)
    ip = np.array([0, 1, 2], dtype=np.int32)
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.Solver(ctx, 2, 2, np.ones(2), ip, np.array([0, 5], dtype=np.int32), 1)     # column out of range
    assert "out of range" in str(e.value)
    s = pkg.Solver(ctx, 2, 2, np.ones(2), ip, np.array([0, 1], dtype=np.int32), 1)
    with pytest.raises(pkg.CgAmdError) as e:
        s.iterate(1)                                                                   # before set_rhs
    assert e.value.status == 7
    with pytest.raises(pkg.CgAmdError):
        s.spmm_rowmajor(s.vector("x"), s.vector("r"), 8)                               # nRHS must be 16 or 32
    s.close()
    # a matrix that is ALREADY on the device is validated there: bad indices are an error, not an out-of-bounds gather
    ipg, ixg, dag = (pkg.DeviceBuffer(ctx, hostbuf=a) for a in (np.array([0, 2, 4, 6], dtype=np.int32), np.array([0, 1, 1, 7, 0, 2], dtype=np.int32), np.ones(6)))
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.Solver(ctx, 3, 6, dag, ipg, ixg, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
    assert e.value.status == 1 and "column index out of range" in str(e.value)
    ipb = pkg.DeviceBuffer(ctx, hostbuf=np.array([0, 4, 2, 6], dtype=np.int32))
    ixok = pkg.DeviceBuffer(ctx, hostbuf=np.array([0, 1, 1, 2, 0, 2], dtype=np.int32))
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.Solver(ctx, 3, 6, dag, ipb, ixok, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
    assert "monotone" in str(e.value)
    ipc = pkg.DeviceBuffer(ctx, hostbuf=np.array([0, 2, 4, 5], dtype=np.int32))
    with pytest.raises(pkg.CgAmdError) as e:
        pkg.Solver(ctx, 3, 6, dag, ipc, ixok, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
    assert "aPointers[size] != nonZeros" in str(e.value)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("n,avg,long_row", [(3000, 24, None), (3001, 40, None), (2500, 90, None), (4000, 30, (777, 3500)),
                                            (700, 260, None)])
def test_solver_spmv_and_cg_dense_rows(pkg, gpu, dtype, n, avg, long_row):
    """rows too dense for a 256-row LDS slice: the solver's plan picks the chunked row-block kernel (2, 4 or 8 lanes per
    row) or, when even a 32-row chunk is too large (one very long row, 260 entries per row), the generic kernel.  SpMV and
    the fused d.q against the oracle, then a short CG on a diagonally dominant system with the same pattern."""
    import scipy.sparse as sp
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n + avg)
    indptr, indices, data = rand_csr(rng, n, avg, dtype, empty_rows=True, long_row=long_row)
    x = rand_vec(rng, n, dtype)
    want = cg_oracle.spmv(indptr, indices, data, x, mode=cg_oracle.MODE_SEQUENTIAL)
    s = pkg.Solver(ctx, n, len(indices), data, indptr, indices, 1)
    xb, yb = _buf(pkg, ctx, x), _buf(pkg, ctx, np.full(n, 3, dtype=dtype))
    for fused in (False, True):
        s.spmv(xb, yb, fused_dot=fused)
        ctx.synchronize()
        got = yb.get()
        scale = np.abs(sp.csr_matrix((np.abs(data), indices, indptr), shape=(n, n))) @ np.abs(x) + 1e-30
        assert np.max(np.abs(got - want) / scale) < RTOL[np.dtype(dtype)], fused
    s.close()
    # CG: A = |pattern| made symmetric and diagonally dominant (complex: complex-symmetric, as the recurrence expects)
    P = sp.csr_matrix((np.abs(data).astype(np.float64) + 0.1, indices, indptr), shape=(n, n))
    P = P + P.T
    cplx = np.dtype(dtype).kind == "c"
    A = sp.csr_matrix(P * ((1.0 + 0.05j) if cplx else 1.0) + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 1.0))
    A.sort_indices()
    ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
    b = rand_vec(rng, n, dtype)
    wide = np.complex128 if cplx else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=8, mode=cg_oracle.MODE_SEQUENTIAL)
    s = pkg.Solver(ctx, n, len(ix), da, ip, ix, 1)
    xg, hg = s.solve(b, None, 8)
    s.close()
    single = np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64))
    if long_row is not None:
        return      # one row/column of 3500 entries: CG amplifies rounding by 1e5 within 8 iterations -- SpMV check only
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > (1e-4 if single else 1e-9)
    assert np.max(np.abs(hg[keep, 0] - ho[keep, 0]) / np.abs(ho[keep, 0])) < (1e-4 if single else 1e-10)
    assert np.linalg.norm(xg - xo) / np.linalg.norm(xo) < (1e-4 if single else 1e-9)
