"""One-byte column codes of the single-RHS row-block SpMV (include/cgamd.h: cgamd_solver_index_codes): the kernel reads
aCols[j] as row + dict[code[j]].  The bar is bit-identity with the kernel that reads aCols (reference kernel/real/spmv.cl:21-27,
kernel/complex/spmv.cl:24-33) -- the column is rebuilt exactly, so SpMV and whole CG histories must not move by a bit -- and
parity with the oracle as everywhere else."""
import numpy as np
import pytest

from conftest import rand_csr, rand_vec

pytestmark = pytest.mark.gpu

DT = {"f32": np.float32, "f64": np.float64, "c64": np.complex64, "c128": np.complex128}


@pytest.fixture
def tuned(pkg):
    lib = pkg._lib.load()
    yield lambda **kv: [pkg._lib.check(lib.cgamd_tune(k.encode(), v)) for k, v in kv.items()]
    lib.cgamd_tune(b"index_codes", 1)
    lib.cgamd_tune(b"index_codes16", 1)
    lib.cgamd_tune(b"dev.value_codes", 1)
    lib.cgamd_tune(b"dev.joint_codes", 1)
    lib.cgamd_tune(b"index_codes_min_mb", 32)
    lib.cgamd_tune(b"resident", 1)


def _solver(pkg, ctx, ip, ix, da, dtype):
    import torch
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tip, tix, tda = t(ip.astype(np.int32)), t(ix.astype(np.int32)), t(da.astype(dtype))
    s = pkg.Solver(ctx, len(ip) - 1, len(ix), tda, tip, tix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
    s._keep = (tip, tix, tda)
    return s


def _run(pkg, ctx, ip, ix, da, dtype, b, iters):
    import torch
    dev = torch.device("cuda", 0)
    s = _solver(pkg, ctx, ip, ix, da, dtype)
    n = len(ip) - 1
    x = torch.from_numpy(np.linspace(-1, 1, n).astype(dtype)).to(dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()             # the solver runs on its own stream
    s.spmv(x, y, fused_dot=True)
    bd = torch.from_numpy(b.astype(dtype)).to(dev)
    torch.cuda.synchronize()
    s.set_rhs(bd, None, on_device=True)
    s.iterate(iters)
    ctx.synchronize()
    out = (s.index_codes, y.cpu().numpy(), s.history().copy(), s.x().copy())
    s.close()
    return out


@pytest.mark.parametrize("dt", ["f64", "f32", "c64", "c128"])
def test_coded_spmv_and_cg_are_bit_identical_to_the_index_kernel(pkg, tuned, dt):
    import cg_numpy
    import cg_oracle
    dtype = DT[dt]
    ctx = pkg.Context(0)
    if dt[0] == "c":
        N = 96
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = cg_numpy.rhsA(N, 12.0).flatten()
    else:
        ip, ix, da = cg_numpy.laplace3d(23, 19, 31)
        b = np.linspace(1.0, 2.0, len(ip) - 1)
    iters = 25
    tuned(resident=0, index_codes=1, index_codes_min_mb=0)
    k1, y1, h1, _ = _run(pkg, ctx, ip, ix, da, dtype, b, iters)
    tuned(index_codes=0)
    k0, y0, h0, _ = _run(pkg, ctx, ip, ix, da, dtype, b, iters)
    assert k0 == 0 and k1 == 7        # 3-D 7-point stencil / P1 triangles on a structured grid: 7 offsets each
    assert np.array_equal(y1, y0)
    assert np.array_equal(h1, h0)
    # and the oracle, as for every other path (tolerances of DESIGN.md section 2)
    xo, ho = cg_oracle.cg(ip, ix, da.astype(dtype), b.astype(dtype), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    rel = np.abs(h1[:, 0] - ho[:, 0]) / np.abs(ho[:, 0])
    assert rel.max() < (1e-10 if dt in ("f64", "c128") else 2e-4), rel.max()
    ctx.close()


def test_irregular_matrices(pkg, tuned):
    """ragged rows, empty rows, unsorted columns and duplicates within a band (<= 256 offsets: coded); scattered columns
    (more offsets than the dictionary holds: the handle keeps aCols)"""
    import torch
    rng = np.random.default_rng(5)
    ctx = pkg.Context(0)
    n = 5000
    rows = []
    for i in range(n):
        k = 0 if i % 97 == 0 else int(rng.integers(1, 12))
        offs = rng.integers(-100, 101, k)
        rows.append(np.clip(i + offs, 0, n - 1))
    ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    ix = np.concatenate(rows).astype(np.int32)
    da = rng.standard_normal(len(ix))
    b = rng.standard_normal(n)
    tuned(resident=0, index_codes=1, index_codes_min_mb=0)
    k1, y1, h1, _ = _run(pkg, ctx, ip, ix, da, np.float64, b, 6)
    tuned(index_codes=0)
    k0, y0, h0, _ = _run(pkg, ctx, ip, ix, da, np.float64, b, 6)
    assert 150 < k1 <= 256 and k0 == 0       # clipping at the ends adds a few offsets beyond the 201 of the band
    assert np.array_equal(y1, y0) and np.array_equal(h1, h0, equal_nan=True)
    import scipy.sparse as sp
    A = sp.csr_matrix((da, ix, ip), shape=(n, n))
    assert np.allclose(y1, A @ np.linspace(-1, 1, n), rtol=1e-12, atol=1e-12)
    # scattered columns: too many offsets for the dictionary; a system this small still gets 16-bit block-relative columns
    # (every row block spans fewer than 65 536 columns), one whose row blocks span more keeps aCols
    ip2, ix2, da2 = rand_csr(rng, 4000, 6, np.float64, empty_rows=True)
    tuned(index_codes=1)
    k2, y2, _, _ = _run(pkg, ctx, ip2, ix2, da2, np.float64, rng.standard_normal(4000), 3)
    A2 = sp.csr_matrix((da2, ix2, ip2), shape=(4000, 4000))
    assert k2 == 65536 and np.allclose(y2, A2 @ np.linspace(-1, 1, 4000), rtol=1e-12, atol=1e-12)
    tuned(index_codes16=0)
    k3, y3, _, _ = _run(pkg, ctx, ip2, ix2, da2, np.float64, rng.standard_normal(4000), 3)
    assert k3 == 0 and np.array_equal(y3, y2)
    tuned(index_codes16=1)
    ip4, ix4, da4 = rand_csr(rng, 90_000, 6, np.float64)           # columns anywhere in 90 000: not codable at all
    k4, y4, _, _ = _run(pkg, ctx, ip4, ix4, da4, np.float64, rng.standard_normal(90_000), 3)
    A4 = sp.csr_matrix((da4, ix4, ip4), shape=(90_000, 90_000))
    assert k4 == 0 and np.allclose(y4, A4 @ np.linspace(-1, 1, 90_000), rtol=1e-12, atol=1e-12)
    ctx.close()


def test_default_gate_and_reload(pkg, tuned):
    """default: matrices below `index_codes_min_mb` (32 MB: the sizes the resident and two-launch loops serve) keep aCols;
    multi-RHS handles always do; reload_matrix with a new pattern re-codes"""
    import torch
    import cg_numpy
    ctx = pkg.Context(0)
    ip, ix, da = cg_numpy.laplace3d(20, 20, 20)
    s = _solver(pkg, ctx, ip, ix, da, np.float64)
    assert s.index_codes == 0
    s.close()
    tuned(index_codes_min_mb=0)
    n = len(ip) - 1
    s2 = pkg.Solver(ctx, n, len(ix), da, ip, ix, 2, dtype=np.float64)
    assert s2.index_codes == 0
    s2.close()
    # host matrix, owned by the handle: reload with another pattern of the same size / nnz
    tuned(resident=0)
    s3 = pkg.Solver(ctx, n, len(ix), da, ip, ix, 1, dtype=np.float64)
    assert s3.index_codes == 7
    ixr = ix.copy()
    for r in range(n):               # reverse the column order inside every row: same rows, other code sequence
        ixr[ip[r]:ip[r + 1]] = ix[ip[r]:ip[r + 1]][::-1]
        da[ip[r]:ip[r + 1]] = da[ip[r]:ip[r + 1]][::-1]
    b = np.linspace(1.0, 2.0, n)
    s3.set_rhs(b, None)
    s3.iterate(10)
    h_a = s3.history().copy()
    s3.reload_matrix(da, ip, ixr)
    assert s3.index_codes == 7
    s3.set_rhs(b, None)
    s3.iterate(10)
    h_b = s3.history().copy()
    assert np.allclose(h_a, h_b, rtol=1e-12)          # same matrix, row sums in another order
    s3.close()
    ctx.close()


def test_headline_size_uses_codes_and_keeps_the_residual_identity(pkg):
    """N = 10M (BASELINE config 1): default handle is coded; r_k = b - A x_k and delta_k = r_k.r_k hold at full size"""
    import torch
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    nx, ny, nz = 250, 200, 200
    n = nx * ny * nz
    ip, ix, da = pkg.generators.laplace3d(ctx, nx, ny, nz, dtype=np.float64)
    s = pkg.Solver(ctx, n, int(ix.numel()), da, ip, ix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
    assert s.index_codes == 7
    b = torch.full((n,), 5.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()             # the solver runs on its own stream
    s.set_rhs(b, None, on_device=True)
    s.iterate(30)
    h = s.history()
    assert h[0, 0] == 25.0 * n
    x = torch.from_numpy(s.x()).to(dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()             # the solver runs on its own stream
    s.spmv(x, y)
    ctx.synchronize()
    r = b - y
    d = float((r * r).sum())
    assert abs(d - h[-1, 0]) / h[-1, 0] < 1e-9
    # A.1 = 0 in the interior, > 0 on the faces: a shifted stencil read through wrong offsets would break this
    one = torch.ones(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(one)
    torch.cuda.synchronize()
    s.spmv(one, y)
    ctx.synchronize()
    yy = y.view(nz, ny, nx)
    assert float(yy[1:-1, 1:-1, 1:-1].abs().max()) == 0.0 and float(y.min()) >= 0.0 and float(y.max()) == 3.0
    s.close()
    ctx.close()


@pytest.mark.parametrize("dt", ["f64", "c64"])
def test_dense_rows_chunked_kernel(pkg, tuned, dt):
    """27-point stencil: the 256-row slice no longer fits LDS, the chunked row-block kernel (2-8 lanes per row) reads the codes"""
    import scipy.sparse as sp
    m = 22
    t1 = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], [-1, 0, 1], format="csr")
    A = sp.kron(sp.kron(t1, t1, format="csr"), t1, format="csr")
    A = (sp.identity(m ** 3, format="csr") * 27.0 - A * 0.5).tocsr()
    A.sort_indices()
    dtype = DT[dt]
    da = A.data.astype(dtype) * ((1 + 0.25j) if dt[0] == "c" else 1)
    ip, ix = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    b = np.linspace(1.0, 2.0, m ** 3)
    ctx = pkg.Context(0)
    tuned(resident=0, index_codes=1, index_codes_min_mb=0)
    k1, y1, h1, x1 = _run(pkg, ctx, ip, ix, da, dtype, b, 12)
    tuned(index_codes=0)
    k0, y0, h0, x0 = _run(pkg, ctx, ip, ix, da, dtype, b, 12)
    assert k0 == 0 and k1 == 27
    assert np.array_equal(y1, y0) and np.array_equal(h1, h0) and np.array_equal(x1, x0)
    ref = sp.csr_matrix((da.astype(np.complex128 if dt[0] == "c" else np.float64), ix, ip)) @ np.linspace(-1, 1, m ** 3).astype(dtype)
    assert np.allclose(y1, ref, rtol=1e-12 if dt == "f64" else 2e-5, atol=1e-12 if dt == "f64" else 1e-4)
    ctx.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
@pytest.mark.parametrize("per_row", [7, 60])
def test_block_relative_16bit_codes_are_exact(pkg, gpu, tuned, dtype, per_row):
    """matrices with more than 256 distinct (column - row) offsets -- unstructured patterns, what Matrix-Market files hold (reference
    main.c:20-33) -- whose 256-row blocks span fewer than 65 536 columns: the SpMV reads 16-bit columns relative to the block's first
    column (cgamd_solver_index_codes() == 65536), also in the chunked kernel for denser rows, and rebuilds the very same column:
    products and whole residual histories are bit-identical to the aCols kernel's."""
    import scipy.sparse as sp
    import torch
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(per_row)
    n = 30_000
    rows = np.repeat(np.arange(n), per_row)
    cols = rows + rng.integers(-3000, 3001, len(rows))
    cols = np.where(cols < 0, -cols, np.where(cols > n - 1, 2 * (n - 1) - cols, cols))      # reflected at the ends: no hub rows
    P = sp.coo_matrix((rng.uniform(-1.0, -0.5, len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    P = P + P.T
    A = sp.csr_matrix(P + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 1e-2))
    A.sort_indices()
    ip, ix = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    da = A.data.astype(dtype) * ((1 + 0.25j) if np.dtype(dtype).kind == "c" else 1)
    b = np.linspace(1.0, 2.0, n).astype(dtype)
    xs = rand_vec(rng, n, dtype)
    dev = torch.device("cuda", 0)

    def run():
        s = pkg.Solver(ctx, n, len(da), da, ip, ix, 1, dtype=dtype)
        y = torch.empty(n, dtype=pkg.generators.torch_dtype(dtype), device=dev)
        xt = torch.from_numpy(xs).to(dev)
        torch.cuda.synchronize()
        s.spmv(xt, y, fused_dot=True)
        ctx.synchronize()
        s.set_rhs(b, None)
        s.iterate(25)
        out = (s.index_codes, y.cpu().numpy(), s.history().copy(), s.x().copy())
        s.close()
        return out

    tuned(resident=0, index_codes=1, index_codes_min_mb=0)
    k1, y1, h1, x1 = run()
    tuned(index_codes16=0)
    k0, y0, h0, x0 = run()
    tuned(index_codes16=1, resident=1, index_codes_min_mb=32)
    assert k1 == 65536 and k0 == 0
    assert np.array_equal(y1, y0) and np.array_equal(h1, h0, equal_nan=True) and np.array_equal(x1, x0, equal_nan=True)
    ref = A.astype(np.complex128 if np.dtype(dtype).kind == "c" else np.float64) * (((1 + 0.25j) if np.dtype(dtype).kind == "c" else 1)) @ xs
    assert np.max(np.abs(y1 - ref)) < (1e-3 if np.dtype(dtype) == np.complex64 else 1e-9) * np.max(np.abs(ref))


def test_matrix_market_input_reaches_the_coded_kernel(pkg, gpu, tuned, tmp_path):
    """the documented user entry (reference main.c:20-33): a Matrix-Market file whose pattern is no stencil gets an index form too"""
    import scipy.sparse as sp
    ctx, queue, kernels = gpu
    rng = np.random.default_rng(9)
    n = 20_000
    rows = np.repeat(np.arange(n), 5)
    cols = rows + rng.integers(-800, 801, len(rows))
    cols = np.where(cols < 0, -cols, np.where(cols > n - 1, 2 * (n - 1) - cols, cols))      # reflected at the ends: no hub rows
    keep = rows > cols
    L = sp.coo_matrix((rng.uniform(-1.0, -0.5, keep.sum()), (rows[keep], cols[keep])), shape=(n, n)).tocsr().tocoo()
    deg = np.asarray(abs(L + L.T).sum(axis=1)).ravel()
    p = str(tmp_path / "irregular.mtx")
    pkg.mmio.mmwrite(p, n, np.concatenate([L.row, np.arange(n)]), np.concatenate([L.col, np.arange(n)]),
                     np.concatenate([L.data, deg + 0.01]), "real", "symmetric")
    n_, ip, ix, da = pkg.mmio.mmread(p)
    tuned(resident=0, index_codes_min_mb=0)
    try:
        s = pkg.Solver(ctx, n_, len(da), da, ip, ix, 1, dtype=np.float64)
        assert s.index_codes == 65536
        b = np.linspace(1.0, 2.0, n_) * np.where(np.arange(n_) % 3 == 0, -1.0, 1.0)      # (b = 5 is an eigenvector of this matrix)
        x, h = s.solve(b, None, 20)
        s.close()
    finally:
        tuned(resident=1, index_codes_min_mb=32)
    import cg_oracle
    xo, ho = cg_oracle.cg(ip, ix, da, b, n_iterations=20, mode=cg_oracle.MODE_SEQUENTIAL)
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > 1e-8
    assert keep.sum() >= 10 and np.max(np.abs(h[keep, 0] - ho[keep, 0]) / np.abs(ho[keep, 0])) < 1e-10


def _run_values(pkg, ctx, ip, ix, da, dtype, b, iters):
    import torch
    dev = torch.device("cuda", 0)
    s = _solver(pkg, ctx, ip, ix, da, dtype)
    n = len(ip) - 1
    x = torch.from_numpy((np.linspace(-1, 1, n) + (0.3j * np.cos(np.arange(n)) if np.dtype(dtype).kind == "c" else 0)).astype(dtype)).to(dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    s.spmv(x, y, fused_dot=True)
    s.spmv(x, y, fused_dot=False)
    bd = torch.from_numpy(b.astype(dtype)).to(dev)
    torch.cuda.synchronize()
    s.set_rhs(bd, None, on_device=True)
    s.iterate(iters)
    ctx.synchronize()
    out = (s.value_codes, s.index_codes, y.cpu().numpy(), s.history().copy(), s.x().copy(), s.spmv_moved_bytes, s.joint_codes)
    s.close()
    return out


@pytest.mark.parametrize("dt,kind", [("f64", "lap3d"), ("f32", "lap3d"), ("c64", "lap3d"), ("f64", "lap2d"), ("f64", "v256"), ("f64", "v257"),
                                      ("c64", "helm"), ("c128", "lap3d")])
def test_value_codes_change_no_bit(pkg, tuned, dt, kind):
    """One-byte VALUE codes (matrices of at most 256 distinct entries: the constant-coefficient stencils of the reference's test
    systems, clcg.c / helmFE_var.Poisson) on top of the one-byte column codes: the kernel multiplies vdict[vcode[j]] -- the same bits
    as aValues[j] -- so SpMV, fused d.q, residual history and x are bit-identical to the kernel that streams aValues.  Exactly 256
    distinct values are coded, 257 are not; the variable-coefficient FE matrix and complex128 keep aValues."""
    import cg_numpy
    import cg_oracle
    dtype = DT[dt]
    ctx = pkg.Context(0)
    rng = np.random.default_rng(5)
    if kind == "helm":
        N = 96
        rho = 1.0 + 0.3 * rng.random((N - 1, N - 1))
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, rho, 0.15, N, N)
    elif kind == "lap2d":
        ip, ix, da = cg_numpy.poisson2d(150)
    else:
        ip, ix, da = cg_numpy.laplace3d(23, 19, 31)
    n, nnz = len(ip) - 1, len(ix)
    if kind in ("v256", "v257"):
        # the same pattern with 256 / 257 distinct entries: symmetric, diagonally dominant
        import scipy.sparse as sp
        nv = int(kind[1:])
        A = sp.csr_matrix((da, ix, ip), shape=(n, n)).tocoo()
        lo = A.row > A.col
        pool = -(1.0 + np.arange(nv - 1) / 1024.0)              # nv - 1 off-diagonal values + one diagonal value
        w = pool[rng.integers(0, nv - 1, int(lo.sum()))]
        w[: nv - 1] = pool                                       # every one of them occurs
        L = sp.coo_matrix((w, (A.row[lo], A.col[lo])), shape=(n, n))
        M = (L + L.T + sp.identity(n) * 16.0).tocsr()
        M.sort_indices()
        ip, ix, da = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.copy()
        assert len(np.unique(da)) == nv
    if dt[0] == "c" and kind != "helm":
        da = da * (1.0 + 0.25j)                                  # complex symmetric, still few distinct entries
    b = np.linspace(1.0, 2.0, n) * (1 + (0.5j if dt[0] == "c" else 0))
    iters = 25
    tuned(resident=0, resident_wide=0, index_codes=1, index_codes_min_mb=0, **{"dev.value_codes": 1})
    v1, k1, y1, h1, x1, mb1, j1 = _run_values(pkg, ctx, ip, ix, da, dtype, b, iters)
    tuned(**{"dev.joint_codes": 0})
    v2, k2, y2, h2, x2, mb2, j2 = _run_values(pkg, ctx, ip, ix, da, dtype, b, iters)      # two code bytes per non-zero
    tuned(**{"dev.value_codes": 0, "dev.joint_codes": 1})
    v0, k0, y0, h0, x0, mb0, j0 = _run_values(pkg, ctx, ip, ix, da, dtype, b, iters)
    pkg._lib.check(pkg._lib.load().cgamd_tune(b"resident_wide", 1))
    assert v0 == 0 and k0 == k1 and k1 in (5, 7)
    want = {"lap3d": 2, "lap2d": 2, "v256": 256, "v257": 0, "helm": 0}[kind] if dt != "c128" else 0
    assert v1 == want, (v1, want)
    V = np.dtype(dtype).itemsize
    # joint codes: one byte names the (offset, value) pair -- the stencils have as many pairs as offsets, 256 values on 7 offsets too many
    assert j0 == 0 and j2 == 0 and j1 == ({"lap3d": 7, "lap2d": 5}.get(kind, 0) if want else 0), (j1, kind)
    assert mb0 - mb2 == (len(ix) * (V - 1) if want else 0)       # the moved-byte model follows the form that runs
    assert mb0 - mb1 == (len(ix) * V if j1 else len(ix) * (V - 1) if want else 0)
    assert np.array_equal(y1, y0) and np.array_equal(h1, h0) and np.array_equal(x1, x0)
    assert np.array_equal(y2, y0) and np.array_equal(h2, h0) and np.array_equal(x2, x0)
    if kind != "helm":
        wide = np.complex128 if dt[0] == "c" else np.float64
        xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
        live = np.abs(ho[:, 0]) > 1e-6 * np.abs(ho[0, 0])
        rel = (np.abs(h1[:, 0] - ho[:, 0]) / np.abs(ho[:, 0]))[live]
        assert rel.max() < (1e-10 if dt in ("f64", "c128") else 5e-4), rel.max()
    ctx.close()
