"""CPU tests: the oracle (numpy + C restatement) is pinned against golden vectors produced by the
UNMODIFIED reference (oracle/make_golden.py imported helmFE_var.py and ast-extracted Poisson/CG/local_rect
from p_h-PY_C-CL.py in the build container).  Nothing here needs a GPU."""
import numpy as np
import pytest

import cg_numpy
import cg_oracle

ALL = [np.float32, np.float64, np.complex64, np.complex128]


# ---- generators -----------------------------------------------------------------------------------
@pytest.mark.parametrize("N", [4, 8, 16])
def test_helmholtz_generator_matches_reference(golden, N):
    g = golden["generators"]
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    assert np.array_equal(ip, g[f"helm_const_N{N}_indptr"])
    assert np.array_equal(ix, g[f"helm_const_N{N}_indices"])
    # vectorised evaluation may differ from the scalar Python arithmetic in the last bit (k**2 via pow)
    assert np.allclose(da, g[f"helm_const_N{N}_data"], rtol=4e-16, atol=4e-16)
    assert len(da) == 7 * N * N - 8 * N + 2


def test_helmholtz_generator_variable_speed_and_rectangular(golden):
    g = golden["generators"]
    ip, ix, da = cg_numpy.helm_fe_var(12, 9.5, g["helm_var_C"], 0.3, 12, 12)
    assert np.array_equal(ip, g["helm_var_N12_indptr"]) and np.array_equal(ix, g["helm_var_N12_indices"])
    assert np.allclose(da, g["helm_var_N12_data"], rtol=1e-15, atol=1e-15)
    ip, ix, da = cg_numpy.helm_fe_var(20, 7.0, g["helm_rect_C"], 0.2, 10, 7)
    assert np.array_equal(ip, g["helm_rect_indptr"]) and np.array_equal(ix, g["helm_rect_indices"])
    assert np.allclose(da, g["helm_rect_data"], rtol=1e-15, atol=1e-15)
    # complex symmetric, not Hermitian
    import scipy.sparse as sp
    A = sp.csr_matrix((da, ix, ip), shape=(70, 70))
    assert abs(A - A.T).max() == 0 and abs(A - A.getH()).max() > 0


@pytest.mark.parametrize("N", [4, 8, 16])
def test_rhs_generators(golden, N):
    g = golden["generators"]
    assert np.array_equal(cg_numpy.rhsA(N, 12.0), g[f"rhsA_N{N}"])
    assert np.array_equal(cg_numpy.rhsL(N, 12.0), g[f"rhsL_N{N}"])


@pytest.mark.parametrize("N", [3, 5, 8])
def test_poisson_generator_matches_reference(golden, N):
    g = golden["driver_generators"]
    ip, ix, da = cg_numpy.poisson2d(N)
    assert np.array_equal(ip, g[f"poisson{N}_indptr"])
    assert np.array_equal(ix, g[f"poisson{N}_indices"])
    assert np.array_equal(da, g[f"poisson{N}_data"])


def test_laplace3d_structure():
    nx, ny, nz = 5, 4, 3
    ip, ix, da = cg_numpy.laplace3d(nx, ny, nz)
    n = nx * ny * nz
    assert ip[-1] == 7 * n - 2 * (nx * ny + ny * nz + nx * nz)
    import scipy.sparse as sp
    A = sp.csr_matrix((da, ix, ip), shape=(n, n))
    assert abs(A - A.T).max() == 0
    assert np.all(A.diagonal() == 6.0)
    assert np.linalg.eigvalsh(A.toarray()).min() > 0      # SPD
    # headline sizes quoted in SURVEY §8 / BASELINE.md
    assert 7 * 10_000_000 - 2 * (250 * 200 + 200 * 200 + 250 * 200) == 69_720_000


# ---- CG recurrence --------------------------------------------------------------------------------
def test_numpy_restatement_bit_identical_to_reference_cg(golden):
    """cg_fixed == helmFE_var.CG (helmFE_var.py:507-544) bit for bit, every iterate 0..60"""
    g = golden["cg_iterates"]
    ip, ix, da, b, X = g["helm16_indptr"], g["helm16_indices"], g["helm16_data"], g["helm16_b"], g["helm16_X"]
    for k in range(0, 61, 3):
        x, h = cg_numpy.cg_fixed(ip, ix, da, b, maxit=k)
        assert np.array_equal(x, X[k]), k
        assert len(h) == k + 1
    x, _ = cg_numpy.cg_fixed(ip, ix, da, b, x0=g["helm16_warm_x0"], maxit=5)
    assert np.array_equal(x, g["helm16_warm_X5"])
    for k, Xk in zip(g["helm32_ks"], g["helm32_X"]):
        x, _ = cg_numpy.cg_fixed(g["helm32_indptr"], g["helm32_indices"], g["helm32_data"], g["helm32_b"], maxit=int(k))
        assert np.array_equal(x, Xk)
    for k, Xk in zip(g["poisson8_ks"], g["poisson8_X"]):
        x, _ = cg_numpy.cg_fixed(g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"], g["poisson8_b"],
                                 maxit=int(k), dtype=complex)
        assert np.array_equal(x, Xk)


def test_numpy_tol_variant_matches_driver_cg(golden):
    g = golden["cg_iterates"]
    x, it = cg_numpy.cg_tol(g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"],
                            g["poisson8_b"].astype(complex), tol=1e-8)
    assert np.array_equal(x, g["poisson8_tol1e-8_x"]) and it > 5


@pytest.mark.parametrize("mode", [cg_oracle.MODE_REFERENCE_ORDER, cg_oracle.MODE_SEQUENTIAL])
def test_c_oracle_vs_reference_iterates_c128(golden, mode):
    """the C restatement (clcg.c:250-430 structure, kernel/*.cl arithmetic) against reference iterates;
    tolerance: rounding only, amplified by CG -- 1e-12 early, 1e-9 by iteration 40"""
    g = golden["cg_iterates"]
    ip, ix, da, b, X = g["helm16_indptr"], g["helm16_indices"], g["helm16_data"], g["helm16_b"], g["helm16_X"]
    for k, tol in ((0, 0), (1, 1e-14), (5, 1e-13), (20, 1e-12), (40, 1e-9)):
        x, h = cg_oracle.cg(ip, ix, da, b, n_iterations=k, mode=mode)
        rel = np.linalg.norm(x - X[k]) / max(np.linalg.norm(X[k]), 1e-300)
        assert rel <= tol, (k, rel)
    _, hn = cg_numpy.cg_fixed(ip, ix, da, b, maxit=40)
    _, hc = cg_oracle.cg(ip, ix, da, b, n_iterations=40, mode=mode)
    assert np.max(np.abs(hc[:, 0] - hn) / np.abs(hn)) < 1e-9


def test_c_oracle_real_and_multi_rhs(golden):
    g = golden["cg_iterates"]
    ip, ix, da = g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"]
    for k, Xk in zip(g["poisson8_ks"], g["poisson8_X"]):
        x, _ = cg_oracle.cg(ip, ix, da, g["poisson8_b"], n_iterations=int(k))
        assert np.allclose(x, Xk.real, rtol=1e-9, atol=1e-12)
    B = g["poisson8_multi_B"]
    x, h = cg_oracle.cg(ip, ix, da, B.reshape(-1), nrhs=3, n_iterations=3)
    xs = np.stack([cg_numpy.cg_fixed(ip, ix, da, B[r], maxit=3)[0] for r in range(3)])
    assert np.allclose(x.reshape(3, 64), xs, rtol=1e-10)
    assert h.shape == (4, 3)
    # RHS-major layout: scaling b scales x (linearity of each independent solve)
    assert np.allclose(x.reshape(3, 64)[1], 2 * x.reshape(3, 64)[0], rtol=1e-10)


@pytest.mark.parametrize("dtype", ALL)
def test_c_oracle_precisions_track_fp64(golden, dtype):
    """measured basis of the parity tolerances: c64/f32 follow the fp64 history to ~1e-5
    while delta_k/delta_0 > 1e-4, and diverge near convergence"""
    g = golden["cg_iterates"]
    if np.dtype(dtype).kind == "c":
        ip, ix, da, b = g["helm32_indptr"], g["helm32_indices"], g["helm32_data"], g["helm32_b"]
        wide = np.complex128
    else:
        ip, ix, da = cg_numpy.poisson2d(24)
        b, wide = np.linspace(1.0, 2.0, 576), np.float64
    _, h = cg_oracle.cg(ip, ix, da.astype(dtype), b.astype(dtype), n_iterations=30)
    _, h64 = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=30)
    keep = np.abs(h64[:, 0]) / np.abs(h64[0, 0]) > 1e-4
    tol = 1e-4 if np.dtype(dtype).itemsize <= 8 and np.dtype(dtype) != np.float64 else 1e-10
    assert np.max(np.abs(h[keep, 0] - h64[keep, 0]) / np.abs(h64[keep, 0])) < tol


# ---- single kernels of the oracle --------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ALL)
def test_oracle_kernels_against_numpy(dtype):
    from conftest import rand_csr, rand_vec
    rng = np.random.default_rng(3)
    n, nrhs = 300, 2
    ip, ix, da = rand_csr(rng, n, 9, dtype, empty_rows=True)
    x, y = rand_vec(rng, n * nrhs, dtype), rand_vec(rng, n * nrhs, dtype)
    a = rand_vec(rng, nrhs, dtype)
    import scipy.sparse as sp
    A = sp.csr_matrix((da, ix, ip), shape=(n, n))
    want = np.concatenate([A @ x[r * n:(r + 1) * n] for r in range(nrhs)])
    tol = 1e-4 if np.dtype(dtype).itemsize <= 8 and np.dtype(dtype) != np.float64 else 1e-12
    for mode in (0, 1):
        assert np.allclose(cg_oracle.spmv(ip, ix, da, x, nrhs=nrhs, mode=mode), want, rtol=tol, atol=tol)
        d = cg_oracle.vdot(x, y, nrhs=nrhs, mode=mode)
        wd = [np.dot(x[r * n:(r + 1) * n], y[r * n:(r + 1) * n]) for r in range(nrhs)]   # unconjugated
        assert np.allclose(d, wd, rtol=tol * 10, atol=tol * 10)
    X, Y = x.reshape(nrhs, n), y.reshape(nrhs, n)
    assert np.allclose(cg_oracle.axpy(x, y, a, 1, nrhs=nrhs).reshape(nrhs, n), Y + a[:, None] * X, rtol=tol)
    assert np.allclose(cg_oracle.axpy(x, y, a, 0, nrhs=nrhs).reshape(nrhs, n), Y - a[:, None] * X, rtol=tol)
    assert np.allclose(cg_oracle.aypx(x, y, a, nrhs=nrhs).reshape(nrhs, n), a[:, None] * Y + X, rtol=tol)
    assert np.array_equal(cg_oracle.sub(x, y, nrhs=nrhs), x - y)


def test_oracle_empty_and_tiny():
    ip = np.array([0, 0, 1, 1], dtype=np.int32)
    ix = np.array([2], dtype=np.int32)
    da = np.array([3.0])
    y = cg_oracle.spmv(ip, ix, da, np.array([1.0, 2.0, 5.0]))
    assert np.array_equal(y, [0.0, 15.0, 0.0])
    assert cg_oracle.vdot(np.array([2.0]), np.array([4.0]))[0] == 8.0


def test_byte_model_matches_baseline_table():
    """SURVEY §8d / BASELINE.md §2 figures"""
    assert cg_numpy.spmv_bytes(10_000_000, 69_720_000, 8) == 1_036_640_004
    assert cg_numpy.cg_iter_bytes(10_000_000, 69_720_000, 8) == 1_996_640_004
    assert cg_numpy.cg_iter_bytes(10_000_000, 69_720_000, 8, fused=True) == 1_756_640_004
    assert round(cg_numpy.spmv_bytes(1_000_000, 4_996_000, 8) / 1e6, 2) == 79.95
    assert round(cg_numpy.spmv_bytes(1_000_000, 4_996_000, 8, nrhs=32) / 1e6, 2) == 575.95


def test_numpy_pcg_bit_identical_to_reference_pcg(golden):
    """oracle pcg_diag == the unmodified reference PCG with a diagonal CSR M (helmFE_var.py:546-586): iterates,
    stopping iteration and solution, complex (Helmholtz, M = 1/diag) and real (shifted Poisson)."""
    g = golden["pcg_iterates"]
    for name, tol_key, tol in (("helm16", "jacobi_tol1e-6", 1e-6), ("shifted_poisson8", "jacobi_tol1e-10", 1e-10)):
        ip, ix, da, b, m = (g[f"{name}_{k}"] for k in ("indptr", "indices", "data", "b", "m"))
        X = g[f"{name}_jacobi_X"]
        for k in range(1, X.shape[0] + 1):
            x, i = cg_numpy.pcg_diag(ip, ix, da, b, m, tol=0.0, maxit=k)
            assert i == k - 1
            assert np.array_equal(x, X[k - 1]), (name, k)
        x, i = cg_numpy.pcg_diag(ip, ix, da, b, m, tol=tol, maxit=1000)
        assert i == int(g[f"{name}_{tol_key}_i"])
        assert np.array_equal(x, g[f"{name}_{tol_key}_x"])
    ip, ix, da, b = (g[f"helm16_{k}"] for k in ("indptr", "indices", "data", "b"))
    x, i = cg_numpy.pcg_diag(ip, ix, da, b, None, tol=1e-6, maxit=1000)
    assert i == int(g["helm16_none_tol1e-6_i"]) and np.array_equal(x, g["helm16_none_tol1e-6_x"])
    # history bookkeeping used by the GPU parity tests: entry k is r_k.r_k
    x, i, h = cg_numpy.pcg_diag(ip, ix, da, b, g["helm16_m"], tol=0.0, maxit=7, history=True)
    assert h.shape == (8,) and i == 6
