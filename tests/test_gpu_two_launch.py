"""GPU tests of the two-launch iteration for small systems (spmv.hip "Two-launch iteration"): the SpMV launch computes
beta and d = beta d + r on the fly (reference aypx, clcg.c:415, moved to the head of the next iteration), the second launch
alpha, x += alpha d, r -= alpha q and the r.r partials (clcg.c:317-374).  It must reproduce the three- and four-launch loops
BIT FOR BIT (same operations per element in the same order; the only change is where they run), for every value type,
single and multiple right-hand sides, graph replay and plain launches, and any split of the iteration count (the search
direction ping-pongs between two buffers, so odd splits exercise both parities)."""
import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import ALL_DTYPES, rand_vec

pytestmark = pytest.mark.gpu


def _system(dtype):
    if np.dtype(dtype).kind == "c":
        N = 32
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        return ip, ix, da.astype(dtype)
    ip, ix, da = cg_numpy.poisson2d(41)          # 1681 rows: 7 row blocks, the last one partial
    return ip, ix, da.astype(dtype)


def _run(pkg, ctx, lib, cfg, ip, ix, da, B, X0, nrhs, flags, splits):
    for k, v in cfg.items():
        pkg._lib.check(lib.cgamd_tune(k.encode(), v))
    try:
        s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, nrhs, flags=flags)
        s.set_rhs(B, X0)
        for n in splits:
            s.iterate(n)
        out = s.x(), s.history()
        # a second solve on the same handle (captured graphs are replayed from iteration 0 again)
        s.set_rhs(B, None)
        s.iterate(sum(splits))
        out = out + (s.x(), s.history())
        s.close()
        return out
    finally:
        for k in cfg:
            pkg._lib.check(lib.cgamd_tune(k.encode(), {"dev.spmm_wide_max": -1, "dev.no_fold_alpha": 0}.get(k, 1)))


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("nrhs", [1, 3, 9])
@pytest.mark.parametrize("graph", [True, False])
def test_two_launch_loop_is_bit_identical_to_three_and_four_launch_loops(pkg, gpu, dtype, nrhs, graph):
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = _system(dtype)
    n = len(ip) - 1
    rng = np.random.default_rng(nrhs)
    B = np.concatenate([(r + 1) * 5.0 + rand_vec(rng, n, dtype) for r in range(nrhs)]).astype(dtype)
    X0 = (0.1 * rand_vec(rng, n * nrhs, dtype)).astype(dtype)
    flags = 0 if graph else pkg._lib.NO_GRAPH
    splits = (3, 1, 16, 5, 8)                         # odd and even entry parities, graph of 8 + singles
    two = _run(pkg, ctx, lib, {"two_launch": 1}, ip, ix, da, B, X0, nrhs, flags, splits)
    three = _run(pkg, ctx, lib, {"two_launch": 0}, ip, ix, da, B, X0, nrhs, flags, splits)
    four = _run(pkg, ctx, lib, {"two_launch": 0, "dev.no_fold_alpha": 1}, ip, ix, da, B, X0, nrhs, flags, splits)
    # multi-RHS: the grouped SpMM kernel (one work-group walks the right-hand sides) instead of one work-group per RHS
    grouped = _run(pkg, ctx, lib, {"dev.spmm_wide_max": 0}, ip, ix, da, B, X0, nrhs, flags, splits)
    for a, b_, c, d_ in zip(two, three, four, grouped):
        assert a.shape == b_.shape == c.shape == d_.shape
        assert np.array_equal(a, b_, equal_nan=True), "two-launch vs three-launch loop differ"
        assert np.array_equal(a, c, equal_nan=True), "two-launch vs four-launch loop differ"
        assert np.array_equal(a, d_, equal_nan=True), "wide vs grouped multi-RHS SpMM differ"
    # and it is the reference recurrence: against the oracle at the stated tolerances
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), B.astype(wide), x0=X0.astype(wide), nrhs=nrhs, n_iterations=sum(splits),
                          mode=cg_oracle.MODE_SEQUENTIAL)
    x, h = two[0], two[1]
    # 32-bit types against the fp64 oracle: the (non-monotone) Helmholtz recurrence amplifies fp32 rounding to ~4e-4 within
    # 20 iterations; tests/test_gpu_refprec.py holds these types to the oracle run in their own precision
    tol = 1e-10 if np.dtype(dtype).itemsize // (2 if np.dtype(dtype).kind == "c" else 1) == 8 else 2e-3
    live = np.abs(ho) > 1e-4 * np.abs(ho[0])
    k = min(20, len(ho))
    assert np.max((np.abs(h - ho) / np.abs(ho))[:k][live[:k]]) < tol


def test_two_launch_loop_history_is_complete_after_every_call(pkg, gpu):
    """delta / history of the last iteration are written by a tail kernel: history() right after iterate(1) must hold it,
    and the device vectors d / r must be the ones of that iteration (cgamd_solver_vector follows the ping-pong)"""
    import ctypes
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.poisson2d(20)
    n = 400
    b = np.linspace(1.0, 2.0, n)
    xo, ho = cg_oracle.cg(ip, ix, da, b, n_iterations=7, mode=cg_oracle.MODE_SEQUENTIAL)
    s = pkg.Solver(ctx, n, len(ix), da, ip, ix, 1)
    s.set_rhs(b, None)
    for k in range(1, 8):
        s.iterate(1)
        h = s.history()
        assert h.shape == (k + 1, 1) and np.allclose(h[:, 0], ho[:k + 1, 0], rtol=1e-11)
        # d_k = r_(k-1) + beta d_(k-1): check d against the residual recurrence on the host
        dk = np.empty(n)
        pkg._lib.check(lib.cgamd_memcpy_d2h(ctx.handle, pkg._lib.ptr(dk), ctypes.c_void_p(s.vector("d")), dk.nbytes))
        assert np.all(np.isfinite(dk))
    x = s.x()
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-10
    s.close()
