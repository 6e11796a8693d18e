"""GPU parity of the CG recurrence (reference clcg.c:250-430 == cl.py:96-200 == helmFE_var.py:507-544).

Stated tolerances (measured basis there):
  * fp64/c128 build vs fp64 oracle / golden iterates of the unmodified reference helmFE_var.CG:
    delta_k rtol 1e-10 for k <= 50 (while not converged), ||x - x_ref|| / ||x_ref|| <= 1e-9;
  * fp32/c64 build vs fp64 oracle: delta_k rtol 1e-4 while delta_k/delta_0 > 1e-4.
The post-convergence tail is never compared (nobody stops iterating: clcg.c:297; it can reach 0/0).
"""
import numpy as np
import pytest

import cg_numpy
import cg_oracle

pytestmark = pytest.mark.gpu


def _solve(pkg, gpu, indptr, indices, data, b, nrhs, iters, dtype, x0=None, flags=0):
    ctx, queue, kernels = gpu
    n = len(indptr) - 1
    s = pkg.Solver(ctx, n, len(indices), data.astype(dtype), indptr, indices, nrhs, flags=flags)
    x, h = s.solve(np.asarray(b).astype(dtype), None if x0 is None else np.asarray(x0).astype(dtype), iters)
    s.close()
    return x, h


def test_golden_helmholtz16_c128_iterates(pkg, gpu, golden):
    """x_k for k = 0..40 against the iterates captured from the unmodified reference CG (complex128)."""
    g = golden["cg_iterates"]
    ip, ix, da, b, X = g["helm16_indptr"], g["helm16_indices"], g["helm16_data"], g["helm16_b"], g["helm16_X"]
    for k in (0, 1, 2, 5, 10, 20, 30, 40):
        x, h = _solve(pkg, gpu, ip, ix, da, b, 1, k, np.complex128)
        rel = np.linalg.norm(x - X[k]) / max(np.linalg.norm(X[k]), 1e-300)
        assert rel < 1e-9, (k, rel)
        assert h.shape == (k + 1, 1)
    # residual history against the fp64 restatement (bit-identical to the reference CG, tests/test_oracle_golden.py)
    _, hn = cg_numpy.cg_fixed(ip, ix, da, b, maxit=50)
    x, h = _solve(pkg, gpu, ip, ix, da, b, 1, 50, np.complex128)
    k = np.arange(0, 41)
    assert np.max(np.abs(h[k, 0] - hn[k]) / np.abs(hn[k])) < 1e-10


def test_golden_helmholtz16_warm_start(pkg, gpu, golden):
    g = golden["cg_iterates"]
    x, _ = _solve(pkg, gpu, g["helm16_indptr"], g["helm16_indices"], g["helm16_data"], g["helm16_b"], 1, 5,
                  np.complex128, x0=g["helm16_warm_x0"])
    assert np.linalg.norm(x - g["helm16_warm_X5"]) / np.linalg.norm(g["helm16_warm_X5"]) < 1e-9


def test_golden_helmholtz32_selected_iterates(pkg, gpu, golden):
    g = golden["cg_iterates"]
    for k, Xk in zip(g["helm32_ks"], g["helm32_X"]):
        if k > 50:
            continue
        x, _ = _solve(pkg, gpu, g["helm32_indptr"], g["helm32_indices"], g["helm32_data"], g["helm32_b"], 1, int(k), np.complex128)
        assert np.linalg.norm(x - Xk) / np.linalg.norm(Xk) < 1e-9, k


def test_golden_poisson8_real_and_multi_rhs(pkg, gpu, golden):
    g = golden["cg_iterates"]
    ip, ix, da = g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"]
    for k, Xk in zip(g["poisson8_ks"], g["poisson8_X"]):
        x, _ = _solve(pkg, gpu, ip, ix, da, g["poisson8_b"], 1, int(k), np.float64)
        assert np.linalg.norm(x - Xk.real) / np.linalg.norm(Xk.real) < 1e-9, k
    # three right-hand sides b[r] = (r+1)*5 (main.c:41-46), each with its own alpha/beta (clcg.c:317-333)
    B = g["poisson8_multi_B"]
    x, h = _solve(pkg, gpu, ip, ix, da, B.reshape(-1), 3, 10, np.float64)
    want = g["poisson8_multi_X10"].real
    # this system converges exactly within 10 steps for constant b: compare where the reference is finite
    ok = np.isfinite(want).all(axis=1)
    assert np.allclose(x.reshape(3, 64)[ok], want[ok], rtol=1e-8, atol=1e-10)
    assert h.shape == (11, 3)


@pytest.mark.parametrize("dtype,rtol", [(np.complex64, 1e-4), (np.float32, 1e-4), (np.float64, 1e-10), (np.complex128, 1e-10)])
@pytest.mark.parametrize("flags", [0, 4, 2])   # fused+graph (default), reference op structure, no graph
def test_residual_history_vs_oracle(pkg, gpu, golden, dtype, rtol, flags):
    """delta history on Helmholtz N=32 (complex) / Poisson 40x40 (real) vs the fp64 C oracle"""
    if np.dtype(dtype).kind == "c":
        g = golden["cg_iterates"]
        ip, ix, da, b = g["helm32_indptr"], g["helm32_indices"], g["helm32_data"], g["helm32_b"]
        wide = np.complex128
    else:
        ip, ix, da = cg_numpy.poisson2d(40)
        b = np.linspace(1.0, 2.0, 1600)
        wide = np.float64
    iters = 50
    x, h = _solve(pkg, gpu, ip, ix, da, b, 1, iters, dtype, flags=flags)
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > 1e-4
    keep[41:] = False
    assert np.max(np.abs(h[keep, 0] - ho[keep, 0]) / np.abs(ho[keep, 0])) < rtol
    if rtol < 1e-6:
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-8


def test_reference_entry_points(pkg, gpu, golden):
    """CG(...) with the reference's positional signature (cl.py:44) and the C entry cg() (clcg.h:3-5)"""
    ctx, queue, kernels = gpu
    g = golden["cg_iterates"]
    ip, ix = g["helm32_indptr"], g["helm32_indices"]
    a = g["helm32_data"].astype(np.csingle)
    n = len(ip) - 1
    nrhs = 3
    b = np.concatenate([g["helm32_b"] * (r + 1) for r in range(nrhs)]).astype(np.csingle)
    x = np.zeros(n * nrhs, dtype=np.csingle)
    out = pkg.CG(ctx, queue, kernels, n, len(a), a, b, ip, ix, x, nrhs, 30)
    assert out is x
    xo, _ = cg_oracle.cg(ip, ix, g["helm32_data"], b.astype(np.complex128), nrhs=nrhs, n_iterations=30,
                         mode=cg_oracle.MODE_SEQUENTIAL)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 5e-4
    x2 = np.zeros(n * nrhs, dtype=np.csingle)
    pkg.cl.cg(n, len(a), a, b, ip, ix, x2, nrhs, 30, 1)
    assert np.array_equal(x, x2)            # same kernels, same order: bitwise reproducible
    x3 = np.zeros(n * nrhs, dtype=np.csingle)
    pkg.conjugate_gradient_multi_gpu(ctx, queue, kernels, n, len(a), a, b, ip, ix, x3, nrhs, 30, pkg.get_gpu_devices()[0])
    assert np.array_equal(x, x3)


def test_size_below_256_and_odd_sizes(pkg, gpu):
    """the reference warns 'size less than 256 NOT SUPPORTED' (clcg.c:123) and reads out of bounds when
    size % 8 != 0 (spmv.cl:18-19); this build must be correct for any size >= 1"""
    for n in (1, 2, 100, 255, 257, 1001):
        N = max(int(np.sqrt(n)), 1)
        rng = np.random.default_rng(n)
        import scipy.sparse as sp
        M = sp.random(n, n, density=min(1.0, 4.0 / n), random_state=np.random.RandomState(n), format="csr")
        A = sp.csr_matrix(M + M.T + sp.identity(n) * (abs(M).sum() + 1.0))
        b = rng.standard_normal(n)
        iters = min(n, 8)
        x, h = _solve(pkg, gpu, A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, b, 1, iters, np.float64)
        xo, ho = cg_oracle.cg(A.indptr, A.indices, A.data, b, n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
        keep = np.abs(ho[:, 0]) > 1e-12 * np.abs(ho[0, 0])
        assert np.allclose(h[keep, 0], ho[keep, 0], rtol=1e-8), n


def test_long_run_reproducible_across_graph_replays(pkg, gpu):
    """300 iterations in 6 batches.  Launched loops (tuning knob resident = 0): graph-replayed batches are bitwise reproducible run
    to run and equal to plain launches.  Default (490k rows, one right-hand side: the chip-wide resident loop): bitwise
    reproducible run to run, and equal to the launched loops within rounding while the residual is alive."""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    N = 700
    indptr, indices, data = pkg.generators.poisson2d(ctx, N, dtype=np.float64)
    n = N * N
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(11)
    b = torch.rand(n, dtype=torch.float64, device=dev, generator=g)

    def run(flags, resident):
        pkg._lib.check(lib.cgamd_tune(b"resident", resident))
        try:
            s = pkg.Solver(ctx, n, indices.numel(), data, indptr, indices, 1, flags=pkg._lib.MATRIX_ON_DEVICE | flags, dtype=np.float64)
        finally:
            pkg._lib.check(lib.cgamd_tune(b"resident", 1))
        torch.cuda.synchronize()
        s.set_rhs(b, None, on_device=True)
        kind = lib.cgamd_solver_loop_launches(s.handle)
        for _ in range(6):
            s.iterate(50)
        x = s.x(torch.empty(n, dtype=torch.float64, device=dev))
        out = (x.clone(), s.history().copy(), kind)
        s.close()
        return out

    launched = [run(0, 0), run(0, 0), run(pkg._lib.NO_GRAPH, 0)]
    assert all(o[2] >= 2 for o in launched)
    for k in (1, 2):
        assert torch.equal(launched[0][0], launched[k][0]) and np.array_equal(launched[0][1], launched[k][1])
    assert launched[0][1].shape == (301, 1) and np.all(np.isfinite(launched[0][1]))
    wide = [run(0, 1), run(0, 1)]
    assert all(o[2] == 1 for o in wide)
    assert torch.equal(wide[0][0], wide[1][0]) and np.array_equal(wide[0][1], wide[1][1])
    hl, hw = launched[0][1][:, 0], wide[0][1][:, 0]
    live = hl > 1e-6 * hl[0]
    assert np.max(np.abs(hw - hl)[live] / hl[live]) < 1e-9


def test_as_prec_shaped_batched_solve(pkg, gpu, golden):
    """n_my sub-domain residuals, one shared matrix, complex64, fixed iterations
    (p_h-PY_C-CL.py:1918-1937); checked per sub-domain against the fp64 oracle"""
    ctx, queue, kernels = gpu
    d = golden["driver_generators"]
    ip, ix, da = d["local_rect_indptr"], d["local_rect_indices"], d["local_rect_data"]      # reference local_rect()
    n = len(ip) - 1
    rng = np.random.default_rng(2)
    res = [(rng.standard_normal((6, 9)) + 1j * rng.standard_normal((6, 9))) for _ in range(5)]
    assert n == 54
    out = pkg.solve_subdomains(ctx, (ip, ix, da), res, 20)
    assert len(out) == 5 and out[0].shape == (6, 9) and out[0].dtype == np.complex128
    for p in range(5):
        xo, _ = cg_oracle.cg(ip, ix, da, res[p].ravel(), n_iterations=20, mode=cg_oracle.MODE_SEQUENTIAL)
        assert np.linalg.norm(out[p].ravel() - xo) / np.linalg.norm(xo) < 5e-4       # complex64 vs fp64
    # resident-matrix form: same numbers, bit for bit
    s = pkg.Solver(ctx, n, len(ix), da.astype(np.csingle), ip, ix, 5)
    out2 = pkg.solve_subdomains(ctx, (ip, ix, da), res, 20, solver=s)
    s.close()
    assert all(np.array_equal(a, b) for a, b in zip(out, out2))


def test_tolerance_stopping_mode(pkg, gpu, golden):
    """the reference's tol-stopping NumPy CG (p_h-PY_C-CL.py:1338-1369) as a mode of the handle"""
    ctx, queue, kernels = gpu
    g = golden["cg_iterates"]
    ip, ix, da, b = g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"], g["poisson8_b"]
    s = pkg.Solver(ctx, 64, len(ix), da, ip, ix, 1)
    x, its, h = s.solve_tol(b, tol=1e-8, maxit=200, check_every=4)
    s.close()
    want = g["poisson8_tol1e-8_x"].real          # reference run stopped after 23 iterations
    assert its == 23                             # the exact stopping iteration, although the history is read every 4
    assert np.linalg.norm(x - want) / np.linalg.norm(want) < 1e-8
    assert np.sqrt(abs(h[-1, 0])) < 1e-8


def test_jacobi_pcg_against_reference_iterates(pkg, gpu, golden):
    """diagonally preconditioned CG = the reference's PCG(A, b, M) with a diagonal CSR M
    (helmFE_var.py:546-586).  x_k against the iterates of the unmodified reference (complex128 Helmholtz with M = 1/diag,
    real shifted Poisson), r_k.r_k against the oracle restatement (bit-identical to the reference on CPU), the
    tolerance-stopping entry against the reference's (x, i), and plain CG again once the preconditioner is removed."""
    ctx, queue, kernels = gpu
    g = golden["pcg_iterates"]
    for name, dtype in (("helm16", np.complex128), ("shifted_poisson8", np.float64)):
        ip, ix, da, b, m = (g[f"{name}_{k}"] for k in ("indptr", "indices", "data", "b", "m"))
        X = g[f"{name}_jacobi_X"]
        n = len(ip) - 1
        s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, 1)
        s.set_preconditioner(m.astype(dtype))
        for k in (1, 2, 3, 7, 12, X.shape[0]):
            x, h = s.solve(b.astype(dtype), None, k)
            rel = np.linalg.norm(x - X[k - 1]) / np.linalg.norm(X[k - 1])
            assert rel < 1e-9, (name, k, rel)
            _, _, ho = cg_numpy.pcg_diag(ip, ix, da, b, m, tol=0.0, maxit=k, history=True)
            assert h.shape == (k + 1, 1)
            keep = np.abs(ho) / np.abs(ho[0]) > 1e-8           # reduction-order noise only while not converged
            assert np.max(np.abs(h[keep, 0] - ho[keep]) / np.abs(ho[keep])) < 1e-10, (name, k)
        tol_key, tol = ("jacobi_tol1e-6", 1e-6) if name == "helm16" else ("jacobi_tol1e-10", 1e-10)
        x, i = s.pcg(b.astype(dtype), M=m.astype(dtype), tol=tol, maxit=1000, check_every=5)
        assert i == int(g[f"{name}_{tol_key}_i"])
        want = g[f"{name}_{tol_key}_x"]
        assert np.linalg.norm(x - want) / np.linalg.norm(want) < 1e-8
        # pcg() leaves the handle unpreconditioned: the plain recurrence again
        xo, ho = cg_oracle.cg(ip, ix, da.astype(dtype), b.astype(dtype), n_iterations=10, mode=cg_oracle.MODE_SEQUENTIAL)
        x, h = s.solve(b.astype(dtype), None, 10)
        assert np.max(np.abs(h[:, 0] - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-10
        s.close()
    # M = None through the same entry: the reference's unpreconditioned branch (z = r)
    ip, ix, da, b = (g[f"helm16_{k}"] for k in ("indptr", "indices", "data", "b"))
    s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, 1)
    x, i = s.pcg(b, M=None, tol=1e-6, maxit=1000)
    assert i == int(g["helm16_none_tol1e-6_i"])
    assert np.linalg.norm(x - g["helm16_none_tol1e-6_x"]) / np.linalg.norm(x) < 1e-8
    s.close()


@pytest.mark.parametrize("dtype,rtol", [(np.float32, 1e-4), (np.complex64, 1e-4), (np.float64, 1e-10)])
def test_jacobi_pcg_multi_rhs_and_precisions(pkg, gpu, dtype, rtol):
    """the preconditioned kernels on a larger system (several row blocks, vector + tail paths: n = 61*61), 3 right-hand
    sides sharing one M, every value type, against the oracle run per right-hand side"""
    import scipy.sparse as sp
    ctx, queue, kernels = gpu
    N = 61
    ip, ix, da = cg_numpy.poisson2d(N)
    n = N * N
    shift = np.linspace(0.5, 4.0, n)
    A = sp.csr_matrix((da, ix, ip), shape=(n, n)) + sp.diags(shift)
    A = sp.csr_matrix(A); A.sort_indices()
    ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    cplx = np.dtype(dtype).kind == "c"
    if cplx:
        da = da * (1.0 + 0.05j)
    m = 1.0 / sp.csr_matrix((da, ix, ip), shape=(n, n)).diagonal()
    rng = np.random.default_rng(7)
    B = rng.standard_normal((3, n)) + (1j * rng.standard_normal((3, n)) if cplx else 0)
    iters = 12
    s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, 3)
    s.set_preconditioner(m.astype(dtype))
    x, h = s.solve(B.reshape(-1).astype(dtype), None, iters)
    s.close()
    for r in range(3):
        xo, _, ho = cg_numpy.pcg_diag(ip, ix, da, B[r].astype(complex), m, tol=0.0, maxit=iters, history=True)
        keep = np.abs(ho) / np.abs(ho[0]) > 1e-4
        assert np.max(np.abs(h[keep, r] - ho[keep]) / np.abs(ho[keep])) < rtol, (dtype, r)
        assert np.linalg.norm(x[r * n:(r + 1) * n] - xo) / np.linalg.norm(xo) < max(rtol * 10, 1e-9)
