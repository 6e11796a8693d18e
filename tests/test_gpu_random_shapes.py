"""Randomised shapes through the handle API: tiny and odd sizes (1 ... a few thousand rows, every remainder modulo the 16-byte
pack), empty rows, several right-hand sides, all four value types, host and device matrices -- the solver against the oracle's
sequential recurrence (reference clcg.c:250-430) and SpMV against scipy.  Seeds are fixed: the cases are the same on every run."""
import numpy as np
import pytest

import cg_oracle

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64, np.complex64, np.complex128]


def _spd(rng, n, half_band, dtype, empty_rows):
    """symmetric (complex-symmetric), diagonally dominant band matrix; optionally a few rows with the diagonal only removed
    entirely (rows AND columns), which leaves empty rows -- x stays at x0 there"""
    import scipy.sparse as sp
    A = sp.lil_matrix((n, n), dtype=np.complex128 if np.dtype(dtype).kind == "c" else np.float64)
    for d in range(1, min(half_band, n - 1) + 1):
        v = rng.uniform(-1.0, -0.1, n - d) * (rng.random(n - d) < 0.7)
        if np.dtype(dtype).kind == "c":
            v = v * (1.0 + 0.3j * rng.uniform(-1, 1, n - d))
        A.setdiag(v, d)
    A = sp.csr_matrix(A)
    A = A + A.T
    diag = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    A = sp.csr_matrix(A + sp.diags(diag.astype(A.dtype)))
    if empty_rows and n > 8:
        kill = rng.choice(n, size=max(1, n // 50), replace=False)
        keep = np.ones(n)
        keep[kill] = 0.0
        D = sp.diags(keep)
        A = sp.csr_matrix(D @ A @ D)
        A.eliminate_zeros()
    A.sort_indices()
    return A


def _cases():
    rng = np.random.default_rng(2024)
    sizes = [1, 2, 3, 5, 7, 9, 63, 64, 65, 255, 257, 513, 1023, 1025, 2049, 3001, 4099]
    out = []
    for i, n in enumerate(sizes):
        for j in range(2):
            dtype = DTYPES[(i + 2 * j) % 4]
            nrhs = [1, 2, 3, 5, 9][(i + j) % 5]
            out.append((n, dtype, nrhs, int(rng.integers(1, 6)), bool((i + j) % 3 == 0), bool(j)))
    return out


@pytest.mark.parametrize("n,dtype,nrhs,half_band,empty_rows,on_device", _cases())
def test_random_shapes(pkg, gpu, n, dtype, nrhs, half_band, empty_rows, on_device):
    import torch
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(n * 131 + nrhs)
    A = _spd(rng, n, half_band, dtype, empty_rows)
    ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
    if len(ix) == 0:
        pytest.skip("no stored entries")
    cplx = np.dtype(dtype).kind == "c"
    B = (rng.standard_normal(n * nrhs) + (1j * rng.standard_normal(n * nrhs) if cplx else 0)).astype(dtype)
    X0 = (0.1 * rng.standard_normal(n * nrhs)).astype(dtype)
    if empty_rows:              # rows without entries: r = b there for ever; keep them out of the residual (b = 0 there)
        rows_empty = np.diff(ip) == 0
        for k in range(nrhs):
            B[k * n:(k + 1) * n][rows_empty] = 0
    iters = 10
    dev = torch.device("cuda", 0)
    if on_device:
        keep = [torch.from_numpy(a).to(dev) for a in (da, ip, ix)]
        torch.cuda.synchronize()
        s = pkg.Solver(ctx, n, len(ix), keep[0], keep[1], keep[2], nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
    else:
        s = pkg.Solver(ctx, n, len(ix), da, ip, ix, nrhs)
    # SpMV on the caller's (unpadded) arrays
    Xd = torch.from_numpy(X0).to(dev)
    Yd = torch.empty_like(Xd)
    torch.cuda.synchronize()
    s.spmv(Xd, Yd)
    ctx.synchronize()
    wide = np.complex128 if cplx else np.float64
    single = np.dtype(dtype).itemsize // (2 if cplx else 1) == 4
    ref = np.concatenate([A.astype(wide) @ X0[k * n:(k + 1) * n].astype(wide) for k in range(nrhs)])
    scale = np.abs(ref).max() + 1e-30
    assert np.abs(Yd.cpu().numpy() - ref).max() / scale < (2e-6 if single else 1e-14)
    x, h = s.solve(B, X0, iters)
    s.close()
    if nrhs == 1 and not on_device:
        # the same system through the launched loop on one-byte column codes (default only above 32 MB): same numbers
        lib = pkg._lib.load()
        for k_, v_ in (("resident", 0), ("index_codes_min_mb", 0)):
            pkg._lib.check(lib.cgamd_tune(k_.encode(), v_))
        try:
            s2 = pkg.Solver(ctx, n, len(ix), da, ip, ix, 1)
            coded = s2.index_codes
            x2, h2 = s2.solve(B, X0, iters)
            s2.close()
        finally:
            for k_, v_ in (("resident", 1), ("index_codes_min_mb", 32)):
                pkg._lib.check(lib.cgamd_tune(k_.encode(), v_))
        assert 0 < coded <= 2 * half_band + 1
        fin = np.isfinite(h[:, 0]) & np.isfinite(h2[:, 0])
        assert np.allclose(h2[fin], h[fin], rtol=(1e-3 if np.dtype(dtype).itemsize // (2 if np.dtype(dtype).kind == "c" else 1) == 4 else 1e-9))
    for k in range(nrhs):
        xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), B[k * n:(k + 1) * n].astype(wide), x0=X0[k * n:(k + 1) * n].astype(wide),
                              n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
        ok = np.isfinite(ho[:, 0]) & (np.abs(ho[:, 0]) > (1e-5 if single else 1e-12) * np.abs(ho[0, 0]))
        ok[1:] &= np.cumprod(ok[:-1]).astype(bool)          # only up to the first converged / broken-down iteration
        rel = np.abs(h[ok, k] - ho[ok, 0]) / np.abs(ho[ok, 0])
        assert rel.max() < (5e-4 if single else 1e-9), (k, rel.max())
        if ok.all():
            assert np.linalg.norm(x[k * n:(k + 1) * n] - xo) / (np.linalg.norm(xo) + 1e-30) < (5e-3 if single else 1e-8)
