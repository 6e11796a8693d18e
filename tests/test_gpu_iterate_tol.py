"""Tolerance stop on the device (cgamd_solver_iterate_tol; the reference's `tol` loop, p_h-PY_C-CL.py:1338-1369):
handles whose loop is resident leave the loop inside the launch, in the iteration the reference stops in.  Checked against the
host-driven scheme (history read back every 8 iterations, re-run to the exact count) and against the reference's golden."""
import ctypes

import numpy as np
import pytest

import cg_numpy
from conftest import rand_vec

pytestmark = pytest.mark.gpu


def _solve(pkg, ctx, ip, ix, da, b, tol, maxit, knobs):
    lib = pkg._lib.load()
    for k, v in knobs.items():
        pkg._lib.check(lib.cgamd_tune(k.encode(), v))
    try:
        s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, 1)
        kind = lib.cgamd_solver_loop_launches(s.handle)
        x, its, h = s.solve_tol(b, tol=tol, maxit=maxit)
        done = s.iterations_done()
        s.close()
        return x, its, h, kind, done
    finally:
        for k in knobs:
            pkg._lib.check(lib.cgamd_tune(k.encode(), 1))


@pytest.mark.parametrize("dtype,N,tol", [(np.float64, 40, 1e-8), (np.float32, 40, 1e-3), (np.float64, 128, 1e-6)])
def test_device_stop_equals_host_scheme_bit_for_bit_on_the_one_xcd_loop(pkg, gpu, dtype, N, tol):
    ctx, queue, kernels = gpu
    ip, ix, da = cg_numpy.poisson2d(N)
    n = N * N
    b = (1.0 + rand_vec(np.random.default_rng(N), n, np.float64)).astype(dtype)
    xd, itd, hd, kd, dd = _solve(pkg, ctx, ip, ix, da.astype(dtype), b, tol, 2000, {})
    xh, ith, hh, kh, dh = _solve(pkg, ctx, ip, ix, da.astype(dtype), b, tol, 2000, {"resident": 0})
    assert kd == 0 and kh == 2 and dd == itd
    assert itd == ith and 10 < itd < 2000
    assert np.array_equal(xd, xh) and np.array_equal(hd[:itd + 1], hh[:ith + 1])
    assert np.sqrt(abs(hd[itd, 0])) < tol <= np.sqrt(abs(hd[itd - 1, 0]))


def test_device_stop_on_the_chip_wide_loop_and_reference_golden(pkg, gpu, golden):
    ctx, queue, kernels = gpu
    # the reference's own tolerance run (golden: Poisson(8), tol 1e-8, x of the unmodified NumPy CG)
    g = golden["cg_iterates"]
    ip, ix, da, b = g["poisson8_indptr"], g["poisson8_indices"], g["poisson8_data"], g["poisson8_b"]
    x, its, h, kind, done = _solve(pkg, ctx, ip, ix, da, b, 1e-8, 1000, {})
    assert kind == 0 and np.linalg.norm(x - g["poisson8_tol1e-8_x"].real) / np.linalg.norm(x) < 1e-9
    # 90 000 rows: the chip-wide loop; same stopping iteration as the host scheme on the launched loops, x within rounding
    ip, ix, da = cg_numpy.poisson2d(300)
    b = 1.0 + rand_vec(np.random.default_rng(3), 90000, np.float64)
    xd, itd, hd, kd, dd = _solve(pkg, ctx, ip, ix, da, b, 1e-6, 5000, {})
    xh, ith, hh, kh, dh = _solve(pkg, ctx, ip, ix, da, b, 1e-6, 5000, {"resident": 0})
    assert kd == 1 and kh >= 2 and itd == ith == dd and 50 < itd < 5000
    assert np.linalg.norm(xd - xh) / np.linalg.norm(xh) < 1e-9
    assert np.sqrt(abs(hd[itd, 0])) < 1e-6 <= np.sqrt(abs(hd[itd - 1, 0]))
    # maxit reached first: exactly maxit iterations
    xm, itm, hm, km, dm = _solve(pkg, ctx, ip, ix, da, b, 1e-30, 37, {})
    assert itm == 37 == dm and hm.shape[0] == 38


def test_iterate_tol_argument_and_state_errors(pkg, gpu):
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.poisson2d(40)
    run = ctypes.c_int(0)
    s = pkg.Solver(ctx, 1600, len(ix), da, ip, ix, 1)
    assert lib.cgamd_solver_iterate_tol(s.handle, 10, 1e-6, ctypes.byref(run)) == pkg._lib.ERR_STATE        # no right-hand side yet
    s.set_rhs(np.ones(1600), None)
    assert lib.cgamd_solver_iterate_tol(s.handle, 10, -1.0, ctypes.byref(run)) == pkg._lib.ERR_INVALID
    assert lib.cgamd_solver_iterate_tol(s.handle, 0, 1e-6, ctypes.byref(run)) == 0 and run.value == 0
    assert lib.cgamd_solver_iterate_tol(s.handle, 5, 1e-300, ctypes.byref(run)) == 0 and run.value == 5
    s.close()
    s = pkg.Solver(ctx, 1600, len(ix), da, ip, ix, 1, flags=pkg._lib.NO_GRAPH)          # launched loop: not served on the device
    s.set_rhs(np.ones(1600), None)
    assert lib.cgamd_solver_iterate_tol(s.handle, 10, 1e-6, ctypes.byref(run)) == pkg._lib.ERR_STATE
    x, its, h = s.solve_tol(np.ones(1600), tol=1e-6)                                      # ... the host scheme still works
    assert 0 < its < 1000
    s.close()


def test_tolerance_run_survives_a_resident_launch_that_cannot_start(pkg, gpu):
    """the resident launch gives up untouched (test hook: one work-group too few): iterate_tol reports it, Solver.solve_tol goes on with
    the host-driven scheme on the launched loops and returns what it always returned"""
    ctx, queue, kernels = gpu
    ip, ix, da = cg_numpy.poisson2d(40)
    b = 1.0 + rand_vec(np.random.default_rng(9), 1600, np.float64)
    x0, it0, h0, k0, d0 = _solve(pkg, ctx, ip, ix, da, b, 1e-8, 2000, {"resident": 0})
    lib = pkg._lib.load()
    pkg._lib.check(lib.cgamd_tune(b"resident_claim_ms", 40))
    try:
        x1, it1, h1, k1, d1 = _solve(pkg, ctx, ip, ix, da, b, 1e-8, 2000, {"dev.resident_test_short_grid": 1})
    finally:
        pkg._lib.check(lib.cgamd_tune(b"resident_claim_ms", 200))
        pkg._lib.check(lib.cgamd_tune(b"dev.resident_test_short_grid", 0))
    assert k1 == 0 and it1 == it0 and np.array_equal(x1, x0)


@pytest.mark.parametrize("kind,tol", [("poisson", 1e-6), ("poisson", 1e-2), ("helm", 1e-3)])
def test_host_scheme_lands_on_the_stopping_iteration(pkg, gpu, kind, tol):
    """launched loops: the check interval shrinks as the residual nears the tolerance (Solver._run_to_tol), so the loop ends in the
    stopping iteration itself and the exact re-run stays the exception; either way its and x are those of the per-iteration check"""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    if kind == "poisson":
        ip, ix, da = cg_numpy.poisson2d(90)
        b = 1.0 + rand_vec(np.random.default_rng(90), 8100, np.float64)
    else:
        N = 64
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = cg_numpy.rhsA(N, 12.0).flatten()
    pkg._lib.check(lib.cgamd_tune(b"resident", 0))
    try:
        s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, 1)
        assert lib.cgamd_solver_loop_launches(s.handle) >= 2
        calls = []
        plain = s.set_rhs
        s.set_rhs = lambda *a, **k: (calls.append(1), plain(*a, **k))[1]
        x, its, h = s.solve_tol(b, tol=tol, maxit=3000, check_every=8)
        n_calls = len(calls)
        x1, its1, h1 = s.solve_tol(b, tol=tol, maxit=3000, check_every=1)       # the reference's own cadence: a check per iteration
        s.close()
    finally:
        pkg._lib.check(lib.cgamd_tune(b"resident", 1))
    assert its == its1 and 5 < its < 3000
    assert np.array_equal(x, x1) and np.array_equal(h, h1)
    assert np.sqrt(abs(h[its, 0])) < tol <= np.sqrt(abs(h[its - 1, 0]))
    assert n_calls <= 2
    if kind == "poisson" and tol == 1e-6:
        assert n_calls == 1          # smooth convergence over many chunks: no re-run needed
