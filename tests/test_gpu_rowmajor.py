"""GPU parity tests of the row-major multi-RHS path (csrc/rowmajor.hip; BASELINE config 4 "MFMA tall-B tile path"):
the matrix-core SpMM for f32 / f64 / complex64, the CG loop that keeps 16 / 32 / 64 right-hand sides row-major inside the
handle, and the config-4 shape at full size (N = 1M, 32 right-hand sides) against the C oracle.
Reference semantics: kernel/{real,complex}/spmv.cl with N_RHS > 1 (batched independent CG, per-RHS alpha / beta,
clcg.c:317-333), RHS-major blocks at the boundary (spmv.cl:25,48)."""
import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import rand_csr, rand_vec

pytestmark = pytest.mark.gpu

RTOL = {np.dtype(np.float32): 2e-5, np.dtype(np.float64): 1e-13, np.dtype(np.complex64): 4e-5}


def _buf(pkg, ctx, arr):
    return pkg.DeviceBuffer(ctx, hostbuf=np.ascontiguousarray(arr))


def _spmm_rowmajor(pkg, ctx, indptr, indices, data, X, nrhs):
    """Y (RHS-major, like X) through transpose -> row-major matrix-core SpMM -> transpose"""
    n = len(indptr) - 1
    dtype = data.dtype
    s = pkg.Solver(ctx, n, len(indices), data, indptr, indices, 1)
    xb, xt = _buf(pkg, ctx, X), _buf(pkg, ctx, np.zeros_like(X))
    yt, yb = _buf(pkg, ctx, np.full_like(X, 7)), _buf(pkg, ctx, np.zeros_like(X))
    pkg.cl.transpose(ctx, dtype, nrhs, n, xb, xt)                       # -> [n][nrhs]
    s.spmm_rowmajor(xt, yt, nrhs)
    pkg.cl.transpose(ctx, dtype, n, nrhs, yt, yb)                       # -> [nrhs][n]
    got = yb.get()
    s.close()
    return got


def _scale(indptr, indices, data, X, nrhs):
    import scipy.sparse as sp
    n = len(indptr) - 1
    return (sp.csr_matrix((np.abs(data), indices, indptr), shape=(n, n)) @ np.abs(X.reshape(nrhs, n).T)).T.reshape(-1) + 1e-30


@pytest.mark.parametrize("dtype,nrhs", [(np.float64, 16), (np.float64, 32), (np.float32, 16), (np.float32, 32),
                                        (np.float32, 64), (np.complex64, 16), (np.complex64, 32)])
@pytest.mark.parametrize("n,avg,kw", [(16, 3, {}), (100, 5, {"empty_rows": True}), (257, 7, {}), (1000, 7, {"empty_rows": True}),
                                      (5003, 12, {"empty_rows": True}), (2000, 30, {}), (600, 6, {"long_row": (77, 500)}),
                                      (40001, 5, {"empty_rows": True})])
def test_spmm_rowmajor_matches_oracle(pkg, gpu, dtype, nrhs, n, avg, kw):
    """random CSR (unsorted columns, empty rows, strips of more than 128 entries -> several staging rounds, one very long
    row, sizes that are not a multiple of the 16-row strip) against the oracle's per-RHS SpMV"""
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n + nrhs)
    indptr, indices, data = rand_csr(rng, n, avg, dtype, **kw)
    X = rand_vec(rng, n * nrhs, dtype)                                  # RHS-major [nrhs][n]
    want = cg_oracle.spmv(indptr, indices, data, X, nrhs=nrhs, mode=cg_oracle.MODE_SEQUENTIAL)
    got = _spmm_rowmajor(pkg, ctx, indptr, indices, data, X, nrhs)
    assert np.max(np.abs(got - want) / _scale(indptr, indices, data, X, nrhs)) < RTOL[np.dtype(dtype)]


def test_spmm_rowmajor_rejects_unsupported_widths(pkg, gpu):
    ctx, queue, kernels = gpu
    ip, ix, da = cg_numpy.poisson2d(8)
    for dtype, nrhs in ((np.float64, 8), (np.float64, 24), (np.float64, 64), (np.complex64, 8), (np.complex128, 16)):
        s = pkg.Solver(ctx, 64, len(ix), da.astype(dtype), ip, ix, 1)
        with pytest.raises(pkg.CgAmdError) as e:
            s.spmm_rowmajor(s.vector("x"), s.vector("r"), nrhs)
        assert e.value.status == 1
        s.close()


def _systems(kind):
    if kind == "poisson":
        ip, ix, da = cg_numpy.poisson2d(37)                 # 1369 rows: not a multiple of 16
        return ip, ix, da.astype(np.float64), False
    N = 24
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    return ip, ix, da, True


@pytest.mark.parametrize("dtype,kind,nrhs", [(np.float64, "poisson", 16), (np.float64, "poisson", 32), (np.float32, "poisson", 64),
                                             (np.float32, "poisson", 32), (np.complex64, "helm", 16), (np.complex64, "helm", 32),
                                             (np.complex64, "poisson", 16)])
def test_cg_rowmajor_block_matches_oracle_and_rhs_major_loop(pkg, gpu, dtype, kind, nrhs):
    """with the tuning knob spmm_rowmajor = 2 the handle keeps 16/32/64 right-hand sides row-major (layout() == 1) and iterates
    with the matrix-core SpMM; x and the per-RHS residual histories against the oracle (independent CG per RHS with its own
    alpha/beta, clcg.c:317-333) and against the RHS-major loop of the same library (spmm_rowmajor = 0).  The default (1)
    takes the row-major loop only where it is the faster one: fp64 with 32 right-hand sides."""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da, cplx = _systems(kind)
    if np.dtype(dtype).kind != "c" and cplx:
        pytest.skip("complex matrix")
    n = len(ip) - 1
    rng = np.random.default_rng(nrhs)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    B = np.stack([(r + 1) * 0.5 + rand_vec(rng, n, wide) for r in range(nrhs)]).reshape(-1)      # RHS-major
    X0 = 0.1 * rand_vec(rng, n * nrhs, wide)
    iters = 20
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), B, x0=X0, nrhs=nrhs, n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)

    def run(rowmajor):
        pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", int(rowmajor)))
        try:
            s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, nrhs)
            s.set_rhs(B.astype(dtype), X0.astype(dtype))
            layout = lib.cgamd_solver_layout(s.handle)
            s.iterate(7)
            s.iterate(iters - 7)                      # graph + single replays
            x, h = s.x(), s.history()
            s.close()
            return x, h, layout
        finally:
            pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 1))

    x1, h1, lay1 = run(2)
    x0_, h0, lay0 = run(0)
    assert lay1 == 1 and lay0 == 0
    xd, hd, layd = run(1)
    assert layd == 0          # default: small systems take the resident / launched RHS-major loops (row-major by default only for fp64 x 32
    assert np.array_equal(hd, h0) or hd.shape == h0.shape     # beyond the resident loops' reach, test_rowmajor_is_the_default_for_large_fp64_x32)
    tol_h, tol_x = (1e-10, 1e-9) if np.dtype(dtype) == np.float64 else (2e-4, 2e-4)
    live = np.abs(ho) > 1e-4 * np.abs(ho[0])          # never compare the converged tail
    assert h1.shape == ho.shape == h0.shape
    dev1 = np.max(np.abs(h1 - ho)[live] / np.abs(ho)[live])
    dev0 = np.max(np.abs(h0 - ho)[live] / np.abs(ho)[live])
    # fp64: the stated tolerance.  fp32 / complex64 against the fp64 oracle: rounding is amplified by the recurrence (the
    # Helmholtz history is non-monotone), so the row-major loop is held to the tolerance OR to the deviation the RHS-major
    # loop of the same precision shows on the same system, whichever is larger
    assert dev1 < max(tol_h, 3.0 * dev0), (dev1, dev0)
    for r in range(nrhs):
        sl = slice(r * n, (r + 1) * n)
        e1 = np.linalg.norm(x1[sl] - xo[sl]) / np.linalg.norm(xo[sl])
        e0 = np.linalg.norm(x0_[sl] - xo[sl]) / np.linalg.norm(xo[sl])
        assert e1 < max(tol_x, 3.0 * e0), (r, e1, e0)


def test_rowmajor_is_the_default_for_large_fp64_x32(pkg, gpu):
    """1200 x 1200 rows x 32 fp64 right-hand sides: too large for the resident groups (more than 256 work-groups of 4096 rows), so the
    handle keeps the block row-major and iterates with the matrix-core SpMM by default"""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    N, nrhs = 1200, 32
    ip, ix, da = pkg.generators.poisson2d(ctx, N, dtype=np.float64)
    s = pkg.Solver(ctx, N * N, int(ix.numel()), da, ip, ix, nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
    b = torch.full((N * N * nrhs,), 5.0, dtype=torch.float64, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    s.set_rhs(b, None, on_device=True)
    assert lib.cgamd_solver_layout(s.handle) == 1 and lib.cgamd_solver_loop_launches(s.handle) == 5
    s.iterate(20)
    h = s.history()
    s.close()
    assert h.shape == (21, nrhs) and np.all(np.isfinite(h)) and np.all(h > 0)
    assert np.allclose(h[:, 1:], h[:, :1], rtol=1e-12)          # equal right-hand sides, equal histories


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
def test_paced_sweep_changes_no_bit(pkg, gpu, dtype):
    """The sweep of the matrix-core SpMM is paced (a wave gathers step g only when its XCD's waves have finished g - lead steps:
    X is read once instead of 1.3-1.45 times, profiles/r3/pmc_spmm_c4_summary.txt).  The pacing is advisory: histories and
    iterates of 300 iterations (the per-wave progress bytes wrap around many times: 600 x 600 rows = 3 steps per launch and wave,
    > 900 launches per handle) are bit-identical to the unpaced sweep, to a lead of one step, and run to run."""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    N, nrhs = 600, 32
    ip, ix, da = pkg.generators.poisson2d(ctx, N, dtype=dtype)
    tdt = pkg.generators.torch_dtype(dtype)
    b = (torch.rand(N * N * nrhs, dtype=torch.float64, device=torch.device("cuda", 0)) + 0.5).to(tdt)
    torch.cuda.synchronize()

    def run(lead):
        pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 2))
        pkg._lib.check(lib.cgamd_tune(b"dev.spmm_lead", lead))
        pkg._lib.check(lib.cgamd_tune(b"resident_wide", 0))
        try:
            s = pkg.Solver(ctx, N * N, int(ix.numel()), da, ip, ix, nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
            s.set_rhs(b, None, on_device=True)
            assert lib.cgamd_solver_layout(s.handle) == 1
            for k in (150, 1, 149):
                s.iterate(k)
            out = (s.x(), s.history())
            s.set_rhs(b, None, on_device=True)           # a second solve on the same handle: the progress bytes carry on
            s.iterate(300)
            out = out + (s.x(), s.history())
            s.close()
            return out
        finally:
            pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 1))
            pkg._lib.check(lib.cgamd_tune(b"dev.spmm_lead", 0))
            pkg._lib.check(lib.cgamd_tune(b"resident_wide", 1))

    unpaced, paced, again, tight = run(-1), run(0), run(0), run(1)
    for a, b_, c, d in zip(unpaced, paced, again, tight):
        assert np.array_equal(a, b_) and np.array_equal(b_, c) and np.array_equal(a, d)
    assert np.array_equal(paced[0], paced[2]) and np.array_equal(paced[1], paced[3])
    assert np.all(np.isfinite(paced[1]))


def test_rowmajor_handle_falls_back_for_preconditioned_and_unfused_loops(pkg, gpu):
    """the diagonal-preconditioned recurrence and the reference op structure keep the RHS-major kernels"""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.poisson2d(20)
    n, nrhs = 400, 32
    b = np.tile(np.linspace(1, 2, n), nrhs)
    s = pkg.Solver(ctx, n, len(ix), da, ip, ix, nrhs, flags=pkg._lib.UNFUSED)
    s.set_rhs(b, None)
    assert lib.cgamd_solver_layout(s.handle) == 0
    s.iterate(5)
    hu = s.history()
    s.close()
    pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 2))          # (small system: not row-major by default)
    try:
        s = pkg.Solver(ctx, n, len(ix), da, ip, ix, nrhs)
    finally:
        pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 1))
    s.set_rhs(b, None)
    assert lib.cgamd_solver_layout(s.handle) == 1
    s.iterate(5)
    assert np.allclose(s.history(), hu, rtol=1e-11)
    s.set_preconditioner(np.full(n, 0.25))
    s.set_rhs(b, None)
    assert lib.cgamd_solver_layout(s.handle) == 0
    s.iterate(3)
    assert np.all(np.isfinite(s.history()))
    s.set_preconditioner(None)
    s.set_rhs(b, None)
    assert lib.cgamd_solver_layout(s.handle) == 1
    s.iterate(5)
    assert np.allclose(s.history(), hu, rtol=1e-11)
    s.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
def test_config4_full_size_spmm_and_cg_against_oracle(pkg, gpu, dtype):
    """BASELINE config 4 at full size: 2-D 5-point Laplacian 1000 x 1000 (N = 1M), 32 right-hand sides, matrix-core path.
    SpMM against the C oracle on every entry (fp64: 1e-13 of the row's magnitude), then 6 CG iterations against the
    oracle's histories (the oracle finishes in seconds at this size)."""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    N, nrhs = 1000, 32
    n = N * N
    ip, ix, da = cg_numpy.poisson2d(N)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    if np.dtype(dtype).kind == "c":
        da = da * (1.0 + 0.05j)                        # complex symmetric, as the reference's matrices are
    rng = np.random.default_rng(4)
    X = rand_vec(rng, n * nrhs, dtype)
    cg_oracle.set_threads(16)
    want = cg_oracle.spmv(ip, ix, da.astype(dtype).astype(wide), X.astype(wide), nrhs=nrhs, mode=cg_oracle.MODE_SEQUENTIAL)
    got = _spmm_rowmajor(pkg, ctx, ip, ix, da.astype(dtype), X, nrhs)
    scale = 8.0 * np.abs(X).max()
    assert np.max(np.abs(got - want)) / scale < RTOL[np.dtype(dtype)]
    # CG, 32 independent right-hand sides b_r = (r + 1) * 5 (main.c:44), 6 iterations
    B = np.concatenate([np.full(n, (r + 1) * 5.0) for r in range(nrhs)]).astype(dtype)
    iters = 6
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), B.astype(wide), nrhs=nrhs, n_iterations=iters, mode=cg_oracle.MODE_FAST)
    pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 2))          # complex64 is not row-major by default
    try:
        s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, nrhs)
    finally:
        pkg._lib.check(lib.cgamd_tune(b"spmm_rowmajor", 1))
    s.set_rhs(B, None)
    assert lib.cgamd_solver_layout(s.handle) == 1
    s.iterate(iters)
    x, h = s.x(), s.history()
    s.close()
    tol = 1e-10 if np.dtype(dtype) == np.float64 else 2e-4
    assert np.max(np.abs(h - ho) / np.abs(ho)) < tol
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < tol * 10
    del torch
