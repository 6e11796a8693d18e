"""Resident CG loop (csrc/resident.hip): every iteration of an iterate() call of a small system inside ONE launch.  The loop
reproduces the partial-sum structure of the two-launch loop, and the library is built with -ffp-contract=off, so x, the
residual history and the device scalars must equal the two-launch loop's BIT FOR BIT (tuning knob resident = 0), and both
are held to the oracle as everywhere else.  Reference loop: clcg.c:297-419; its workload of this size: the sub-domain
solves of p_h-PY_C-CL.py:1925-1950 (16k rows x 9 complex64 right-hand sides x 256 iterations)."""
import ctypes

import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import rand_vec

pytestmark = pytest.mark.gpu


def _system(kind):
    if kind == "poisson40":
        ip, ix, da = cg_numpy.poisson2d(40)                     # 1600 rows: 2 members, the second one partial
        return ip, ix, da.astype(np.float64)
    if kind == "poisson128":
        ip, ix, da = cg_numpy.poisson2d(128)                    # 16384 rows: 16 members = half an XCD
        return ip, ix, da.astype(np.float64)
    if kind == "poisson200":
        ip, ix, da = cg_numpy.poisson2d(200)                    # 40000 rows: 40 members > one XCD -> write-through form
        return ip, ix, da.astype(np.float64)
    if kind == "helm128":                                       # the reference's sub-domain matrix shape (as_prec)
        N = 128
        return cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    if kind == "helm256":
        N = 256
        return cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    if kind == "helm24":
        N = 24
        return cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    if kind == "band5":                                         # rows of 1 .. 11 entries
        return _banded_spd(6144, 5, 0.8, 5)
    if kind == "band12":                                        # rows of 1 .. 25 entries: longer than any unroll
        return _banded_spd(4096, 12, 0.4, 12)
    if kind == "scattered":
        return _scattered_spd(20480, 3, 3)
    raise ValueError(kind)


def _banded_spd(n, half_band, keep, seed):
    """symmetric, diagonally dominant, unsorted columns, rows of very different lengths (up to 2 half_band + 1 entries,
    some rows with the diagonal only): exercises the 10 / 12-entry instances and the beyond-the-unroll row walk"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    A = sp.lil_matrix((n, n))
    for d in range(1, half_band + 1):
        v = rng.uniform(-1.0, -0.1, n - d) * (rng.random(n - d) < keep)
        A.setdiag(v, d)
    A = sp.csr_matrix(A)
    A = A + A.T
    A = sp.csr_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0))
    ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()
    for r in range(n):                       # unsorted columns inside the rows
        s, e = ip[r], ip[r + 1]
        perm = rng.permutation(e - s)
        ix[s:e], da[s:e] = ix[s:e][perm], da[s:e][perm]
    return ip, ix, da


def _scattered_spd(n, per_row, seed):
    """symmetric, diagonally dominant, columns anywhere in [0, n): the column range of a row slice is the whole vector --
    too wide for the LDS window, the loop gathers d and r per non-zero"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), per_row)
    cols = rng.integers(0, n, rows.size)
    A = sp.csr_matrix((rng.uniform(-1.0, -0.1, rows.size), (rows, cols)), shape=(n, n))
    A.sum_duplicates()
    A = A + A.T
    A = sp.csr_matrix(A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0))
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.copy()


def _run(pkg, ctx, ip, ix, da, B, X0, nrhs, calls, knobs):
    lib = pkg._lib.load()
    for k, v in knobs.items():
        pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))
    try:
        s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, nrhs)
        s.set_rhs(B, X0)
        kind = lib.cgamd_solver_loop_launches(s.handle)
        for c in calls:
            s.iterate(c)
        out = dict(x=s.x(), h=s.history(), kind=kind)
        for name in ("r", "d"):
            buf = np.empty((len(ip) - 1) * nrhs, dtype=da.dtype)
            pkg._lib.check(lib.cgamd_memcpy_d2h(ctx.handle, pkg._lib.ptr(buf), ctypes.c_void_p(s.vector(name)), buf.nbytes))
            out[name] = buf
        s.close()
        return out
    finally:
        for k in knobs:
            pkg._lib.check(lib.cgamd_tune(k.encode(), {"resident": 1, "resident_min": 8, "dev.resident_window": 1, "resident_wide": 1, "resident_wide_min": 16}.get(k, 0)))


CASES = [
    (np.float64, "poisson40", 1, [25, 9]),
    (np.float32, "poisson40", 3, [30]),
    (np.float64, "poisson128", 1, [40, 24]),
    (np.float64, "poisson128", 5, [32]),
    (np.float32, "poisson128", 2, [32]),
    (np.complex64, "helm24", 4, [20, 20]),
    (np.complex128, "helm24", 3, [20, 13]),         # 16-byte values: two packs per thread
    (np.complex128, "poisson128", 2, [24]),         # (7 entries x 20 bytes per row do not fit LDS at 1024 rows: helm128 in complex128 stays launched)
    (np.complex128, "poisson40", 1, [17, 8]),
    (np.complex64, "helm128", 9, [64]),             # the as_prec shape
    (np.complex64, "helm128", 20, [24]),            # more right-hand sides than groups: groups claim several in turn
    (np.float64, "poisson200", 2, [24]),            # groups wider than an XCD
    (np.float32, "band5", 2, [20]),                 # 12-entry instance (fp32 only)
    (np.float64, "band5", 2, [20]),                 # 8-entry instance + rest of the row from LDS
    (np.complex64, "band5", 1, [20]),
    (np.float64, "band12", 3, [12, 9]),
    (np.complex64, "band12", 1, [16]),
    (np.float64, "scattered", 2, [16]),             # no LDS window: per-non-zero gathers by default
    (np.complex64, "helm256", 1, [24]),             # 65536 rows: the largest group (64 work-groups, 256 d.q partials)
]


@pytest.mark.parametrize("dtype,kind,nrhs,calls", CASES)
def test_resident_loop_is_bit_identical_to_two_launch_loop(pkg, gpu, dtype, kind, nrhs, calls):
    ctx, queue, kernels = gpu
    ip, ix, da = _system(kind)
    if np.dtype(dtype).kind != "c" and np.iscomplexobj(da):
        pytest.skip("complex matrix")
    n = len(ip) - 1
    rng = np.random.default_rng(n + nrhs)
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    B = np.concatenate([(r + 1) * 0.5 + rand_vec(rng, n, wide) for r in range(nrhs)])
    X0 = 0.1 * rand_vec(rng, n * nrhs, wide)
    A = da.astype(dtype)
    # the loop needs the vector launch's "one 16-byte pack per thread" partial structure: the default up to 65536 rows
    base = {"dev.vec_ppt": 1} if n * nrhs > 262144 and n > 65536 else {}
    if n > 32768:
        base["resident"] = 2                 # groups wider than an XCD: the write-through form, not taken by default (slower than launches)
    res = _run(pkg, ctx, ip, ix, A, B.astype(dtype), X0.astype(dtype), nrhs, calls, dict(base))
    two = _run(pkg, ctx, ip, ix, A, B.astype(dtype), X0.astype(dtype), nrhs, calls, dict(base, resident=0))
    assert res["kind"] == 0 and two["kind"] == 2
    for key in ("h", "x", "r", "d"):
        assert np.array_equal(res[key], two[key]), key
    # ... and the write-through (cross-XCD) form of the same kernel
    wt = _run(pkg, ctx, ip, ix, A, B.astype(dtype), X0.astype(dtype), nrhs, calls, dict(base, resident=2))
    assert wt["kind"] == 0
    for key in ("h", "x", "r", "d"):
        assert np.array_equal(wt[key], two[key]), key
    # ... and the form without the LDS window (every non-zero gathers d and r from L2: what irregular patterns get)
    nw = _run(pkg, ctx, ip, ix, A, B.astype(dtype), X0.astype(dtype), nrhs, calls, dict(base, **{"dev.resident_window": 0}))
    if np.dtype(dtype) == np.complex128 and n >= 2048:
        # complex128 runs the one-XCD loop only with the window; without it the chip-wide groups take the handle over (round 3), and
        # the handle's launched loops sum their partials in the members' order: bit-identical to each other on that handle
        assert nw["kind"] == 1
        nl = _run(pkg, ctx, ip, ix, A, B.astype(dtype), X0.astype(dtype), nrhs, calls, dict(base, **{"dev.resident_window": 0, "resident_wide_min": 1 << 20}))
        for key in ("h", "x"):
            assert np.array_equal(nw[key], nl[key]), key
            assert np.allclose(nw[key], two[key], rtol=1e-9, atol=0)
    else:
        assert nw["kind"] == (2 if np.dtype(dtype) == np.complex128 else 0)
        for key in ("h", "x", "r", "d"):
            assert np.array_equal(nw[key], two[key]), key
    # against the oracle (fp64: the north star's 1e-10 on delta_k; lower precisions as in test_gpu_cg.py)
    iters = sum(calls)
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), B, x0=X0, nrhs=nrhs, n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    tol = 1e-10 if np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128)) else 5e-3
    live = np.abs(ho) > 1e-4 * np.abs(ho[0])
    assert res["h"].shape == ho.shape
    # fp32 / complex64 against the fp64 oracle: the recurrence amplifies rounding (the Helmholtz history is non-monotone), so
    # only the first 12 iterations are held to the tolerance; the whole run is bit-equal to the two-launch loop (above), which
    # tests/test_gpu_two_launch.py and tests/test_gpu_refprec.py hold to the oracle at full length
    upto = ho.shape[0] if np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128)) else 13
    assert np.max((np.abs(res["h"] - ho) / np.abs(ho))[:upto][live[:upto]]) < tol


def test_resident_and_two_launch_calls_mix(pkg, gpu):
    """calls shorter than resident_min run the two-launch loop on the same state; any interleaving gives the same bits"""
    ctx, queue, kernels = gpu
    ip, ix, da = _system("poisson128")
    n, nrhs = len(ip) - 1, 2
    rng = np.random.default_rng(3)
    B = rand_vec(rng, n * nrhs, np.float64)
    mixed = _run(pkg, ctx, ip, ix, da, B, None, nrhs, [3, 20, 1, 2, 17, 5], {})
    plain = _run(pkg, ctx, ip, ix, da, B, None, nrhs, [48], {"resident": 0})
    one = _run(pkg, ctx, ip, ix, da, B, None, nrhs, [48], {})
    for key in ("h", "x", "r", "d"):
        assert np.array_equal(mixed[key], plain[key]), key
        assert np.array_equal(one[key], plain[key]), key


def test_resident_loop_does_not_apply(pkg, gpu):
    """sizes / types / flags outside the loop's reach keep the launched loops"""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.poisson2d(200)                    # 40000 rows: a group would span two XCDs
    s = pkg.Solver(ctx, 40000, len(ix), da, ip, ix, 2)      # ... it takes the chip-wide form (one group per right-hand side, at most two rounds)
    assert lib.cgamd_solver_loop_launches(s.handle) == 1
    s.close()
    s = pkg.Solver(ctx, 40000, len(ix), da, ip, ix, 2, flags=pkg._lib.NO_GRAPH)     # plain launches requested
    assert lib.cgamd_solver_loop_launches(s.handle) == 2
    s.close()
    ip, ix, da = cg_numpy.poisson2d(37)                     # 1369 rows: odd, no 16-byte packs per right-hand side ...
    s = pkg.Solver(ctx, 1369, len(ix), da, ip, ix, 2)       # ... the handle carries it with one empty row appended (tests/test_gpu_odd_sizes.py)
    assert lib.cgamd_solver_loop_launches(s.handle) == 0 and s.ld == 1370
    s.close()
    pkg._lib.check(lib.cgamd_tune(b"pad_rows", 0))          # as passed: launched loops
    try:
        s = pkg.Solver(ctx, 1369, len(ix), da, ip, ix, 2)
        assert lib.cgamd_solver_loop_launches(s.handle) != 0 and s.ld == 1369
        s.close()
    finally:
        pkg._lib.check(lib.cgamd_tune(b"pad_rows", 1))
    ip, ix, da = cg_numpy.poisson2d(40)
    s = pkg.Solver(ctx, 1600, len(ix), da, ip, ix, 1, flags=pkg._lib.UNFUSED)
    assert lib.cgamd_solver_loop_launches(s.handle) == 8
    s.close()


def test_long_call_is_cut_into_several_resident_launches(pkg, gpu):
    """iterate(40000): more than one launch's worth of iterations (2^15 per launch); the cuts leave no trace in the bits"""
    ctx, queue, kernels = gpu
    ip, ix, da = _system("poisson40")
    n, nrhs = len(ip) - 1, 2
    rng = np.random.default_rng(1)
    B = rand_vec(rng, n * nrhs, np.float64)
    res = _run(pkg, ctx, ip, ix, da, B, None, nrhs, [40000], {})
    two = _run(pkg, ctx, ip, ix, da, B, None, nrhs, [40000], {"resident": 0})
    assert res["kind"] == 0 and two["kind"] == 2 and res["h"].shape == (40001, nrhs)
    for key in ("h", "x", "r", "d"):
        assert np.array_equal(res[key], two[key], equal_nan=True), key
