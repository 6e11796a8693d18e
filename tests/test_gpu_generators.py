"""Product-side generators of the reference's finite-element Helmholtz matrices (csrc/generators.hip): helmFE_var
(helmFE_var.py:9-331) and local_rect (p_h-PY_C-CL.py:1439-1639), written straight into device CSR.  Checked against the golden
matrices the UNMODIFIED reference produced (tests/golden/generators.npz, driver_generators.npz: pattern exact, values to 1e-15)
and against the CPU restatement at BASELINE config 3's full size (N = 500)."""
import numpy as np
import pytest

import cg_numpy

pytestmark = pytest.mark.gpu


def _host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("N", [4, 8, 16])
def test_helm_fe_var_constant_speed_matches_reference_golden(pkg, gpu, golden, N):
    ctx, queue, kernels = gpu
    g = golden["generators"]
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, np.ones((N - 1, N - 1)), 0.15, dtype=np.complex128)
    assert np.array_equal(_host(ip), g[f"helm_const_N{N}_indptr"])
    assert np.array_equal(_host(ix), g[f"helm_const_N{N}_indices"])
    assert np.allclose(_host(da), g[f"helm_const_N{N}_data"], rtol=1e-15, atol=1e-15)
    ip2, ix2, da2 = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex128)      # C = None: all ones
    assert np.array_equal(_host(da2), _host(da)) and np.array_equal(_host(ix2), _host(ix))


def test_helm_fe_var_variable_speed_and_rectangle_match_reference_golden(pkg, gpu, golden):
    ctx, queue, kernels = gpu
    g = golden["generators"]
    ip, ix, da = pkg.generators.helm_fe_var(ctx, 12, 9.5, g["helm_var_C"], 0.3, 12, 12, dtype=np.complex128)
    assert np.array_equal(_host(ip), g["helm_var_N12_indptr"]) and np.array_equal(_host(ix), g["helm_var_N12_indices"])
    assert np.allclose(_host(da), g["helm_var_N12_data"], rtol=1e-15, atol=1e-15)
    ip, ix, da = pkg.generators.helm_fe_var(ctx, 20, 7.0, g["helm_rect_C"], 0.2, 10, 7, dtype=np.complex128)
    assert np.array_equal(_host(ip), g["helm_rect_indptr"]) and np.array_equal(_host(ix), g["helm_rect_indices"])
    assert np.allclose(_host(da), g["helm_rect_data"], rtol=1e-15, atol=1e-15)
    # complex64: the same values rounded once
    ip, ix, d32 = pkg.generators.helm_fe_var(ctx, 20, 7.0, g["helm_rect_C"], 0.2, 10, 7, dtype=np.complex64)
    assert np.array_equal(_host(d32), g["helm_rect_data"].astype(np.complex64))


def test_local_rect_matches_reference_golden(pkg, gpu, golden):
    ctx, queue, kernels = gpu
    g = golden["driver_generators"]
    N, k, eps, eta, L, Nh, Nv = g["local_rect_params"]
    ip, ix, da = pkg.generators.local_rect(ctx, int(N), k, eps, eta, L, int(Nh), int(Nv), dtype=np.complex128)
    assert np.array_equal(_host(ip), g["local_rect_indptr"]) and np.array_equal(_host(ix), g["local_rect_indices"])
    assert np.allclose(_host(da), g["local_rect_data"], rtol=1e-15, atol=1e-15)


def test_config3_generator_at_full_size_and_errors(pkg, gpu):
    ctx, queue, kernels = gpu
    N = 500
    hp, hx, hd = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex128)
    assert len(hx) == 7 * N * N - 8 * N + 2 == ix.numel()
    assert np.array_equal(_host(ip), hp) and np.array_equal(_host(ix), hx)
    assert np.allclose(_host(da), hd, rtol=1e-15, atol=1e-15)
    # the generated matrix drives the solver like any other device CSR (config 3, complex64)
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
    s = pkg.Solver(ctx, N * N, ix.numel(), da, ip, ix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.complex64)
    b = cg_numpy.rhsA(N, 12.0).flatten().astype(np.complex64)
    s.set_rhs(b, None)
    s.iterate(10)
    h = s.history()
    s.close()
    import cg_oracle
    xo, ho = cg_oracle.cg(hp, hx, hd, b.astype(np.complex128), n_iterations=10, mode=cg_oracle.MODE_FAST)
    assert np.max(np.abs(h[:, 0] - ho[:, 0]) / np.abs(ho[:, 0])) < 2e-4
    with pytest.raises(ValueError):
        pkg.generators.helm_fe_var(ctx, 8, 12.0, np.ones((3, 3)), 0.15)
    with pytest.raises(ValueError):
        pkg.generators.local_rect(ctx, 17, 10.0, 10.0, 10.0, 1.0, 9, 6, dtype=np.float64)
    with pytest.raises(pkg.CgAmdError):
        pkg.generators.helm_fe_var(ctx, 8, 12.0, np.zeros((7, 7)), 0.15)


def test_device_rhs_generators_match_the_reference(pkg, gpu, golden):
    """generators.rhsA / rhsL / rhs (csrc/generators.hip) against what the unmodified reference produced (helmFE_var.py:333-389,
    tests/golden/generators.npz: rhsA_N*, rhsL_N*, rhs_N8 with k = 12): rhsA / rhsL exactly, rhs to 1e-14 relative to its largest
    entry (device sin / cos and the reference's BLAS dot differ in the last bit) -- and the boundary quirk of the reference (its right
    boundary integrates the top boundary's points) is reproduced, not repaired"""
    ctx, queue, kernels = gpu
    g = golden["generators"]
    for N in (4, 8, 16):
        for name, fn in (("rhsA", pkg.generators.rhsA), ("rhsL", pkg.generators.rhsL)):
            want = g[f"{name}_N{N}"].flatten()
            got = fn(ctx, N, 12.0, dtype=np.complex128).cpu().numpy()
            assert np.array_equal(got, want), (name, N)
            got32 = fn(ctx, N, 12.0, dtype=np.complex64).cpu().numpy()
            assert np.array_equal(got32, want.astype(np.complex64))
        assert np.array_equal(pkg.generators.rhsA(ctx, N, 12.0, dtype=np.float64).cpu().numpy(), g[f"rhsA_N{N}"].real.flatten())
    want = g["rhs_N8"].flatten()
    got = pkg.generators.rhs(ctx, 8, 12.0, dtype=np.complex128).cpu().numpy()
    assert np.max(np.abs(got - want)) < 1e-14 * np.max(np.abs(want)), np.max(np.abs(got - want))
    assert np.array_equal(got == 0, want == 0)                      # interior nodes are exactly zero, boundary nodes are not
    # config 3's right-hand side at full size against the oracle's restatement
    import cg_numpy
    got = pkg.generators.rhsA(ctx, 500, 12.0, dtype=np.complex64).cpu().numpy()
    assert np.array_equal(got, cg_numpy.rhsA(500, 12.0).flatten().astype(np.complex64))
    with pytest.raises(pkg._lib.CgAmdError):
        pkg.generators.rhs(ctx, 8, 12.0, dtype=np.float64)
