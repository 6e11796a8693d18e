"""Chip-wide resident loop (csrc/resident.hip, cg_resident_wide_kernel): one group of up to 256 work-groups for a single
right-hand side, the matrix in registers, all iterations of an iterate() call in one launch -- BASELINE configs 2 (1M rows,
2-D 5-point, fp64) and 3 (250k rows, helmFE_var(500), complex64).  It is held to the oracle (fp64: 1e-10 on delta_k, as every
loop), and to the launched loops of the SAME handle bit for bit: a handle this loop can take over runs its launched loops with
the member-blocked order of the partial sums (csrc/reduce_device.h thread_partials), which is the order the members of the
resident group form them in -- so the bits do not depend on how a solve is cut into iterate() calls."""
import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import rand_vec

pytestmark = pytest.mark.gpu


def _run(pkg, ctx, ip, ix, da, b, calls, wide, nrhs=1):
    lib = pkg._lib.load()
    pkg._lib.check(lib.cgamd_tune(b"resident_wide", int(wide)))
    try:
        s = pkg.Solver(ctx, len(ip) - 1, len(ix), da, ip, ix, nrhs)
        s.set_rhs(b, None)
        kind = lib.cgamd_solver_loop_launches(s.handle)
        for c in calls:
            s.iterate(c)
        out = dict(x=s.x(), h=s.history(), kind=kind)
        s.close()
        return out
    finally:
        pkg._lib.check(lib.cgamd_tune(b"resident_wide", 1))


@pytest.mark.parametrize("dtype,kind,calls", [
    (np.float64, "poisson300", [24]),            # 90 000 rows: 44 work-groups of 2048 rows
    (np.float64, "poisson1000", [16, 3, 17]),    # config 2 at full size: 245 work-groups of 4096 rows; a launched call in between
    (np.complex64, "helm500", [24]),             # config 3 at full size: 123 work-groups of 2048 rows, 7 entries per row
    (np.float32, "poisson300", [20, 16]),
    (np.complex128, "helm500", [16, 16]),        # config 3 in the build's wide type: 245 work-groups of 1024 rows
    (np.float64, "ninepoint300", [24]),          # 9 entries per row (bilinear elements): the 10-entry instance
    (np.float64, "lap3d64", [24]),               # 3-D 7-point 64^3: 128 work-groups, windows of 2 x 4096 halo rows
    (np.float64, "band50000", [20]),             # irregular rows (1 .. 7 entries, unsorted, some diagonal-only), last member partial
    (np.complex64, "band50000", [20]),
])
def test_wide_resident_loop_against_oracle_and_launched_loop(pkg, gpu, dtype, kind, calls):
    ctx, queue, kernels = gpu
    if kind == "helm500":
        N = 500
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    elif kind.startswith("band"):
        from test_gpu_resident import _banded_spd
        ip, ix, da = _banded_spd(int(kind[4:]), 3, 0.8, 50)
        assert np.diff(ip).max() <= 7 and np.diff(ip).min() == 1
    elif kind.startswith("lap3d"):
        N = int(kind[5:])
        ip, ix, da = cg_numpy.laplace3d(N, N, N)
    elif kind.startswith("ninepoint"):
        import scipy.sparse as sp
        N = int(kind[9:])
        T1 = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
        A9 = sp.csr_matrix(sp.identity(N * N) * 3.0 - sp.kron(T1, T1) / 3.0)      # 8/3 on the diagonal, -1/3 to the 8 neighbours
        A9.sort_indices()
        ip, ix, da = A9.indptr.astype(np.int32), A9.indices.astype(np.int32), A9.data.copy()
        assert np.diff(ip).max() == 9
    else:
        ip, ix, da = cg_numpy.poisson2d(int(kind[7:]))
    n = len(ip) - 1
    wide_t = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    rng = np.random.default_rng(n)
    b = (1.0 + rand_vec(rng, n, wide_t)).astype(dtype)
    A = da.astype(dtype)
    w = _run(pkg, ctx, ip, ix, A, b, calls, True)
    l = _run(pkg, ctx, ip, ix, A, b, calls, False)
    assert w["kind"] == 1 and l["kind"] >= 2
    iters = sum(calls)
    cg_oracle.set_threads(16)
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide_t), b.astype(wide_t), n_iterations=iters, mode=cg_oracle.MODE_FAST)
    tol = 1e-10 if np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128)) else 2e-3
    assert w["h"].shape == ho.shape
    wide64 = np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128))
    upto = ho.shape[0] if wide64 else 13
    assert np.max((np.abs(w["h"][:, 0] - ho[:, 0]) / np.abs(ho[:, 0]))[:upto]) < tol
    # the launched loop of the same precision: same recurrence, different summation grouping
    assert np.max((np.abs(w["h"] - l["h"]) / np.abs(l["h"]))[:upto]) < (1e-11 if wide64 else 2e-3)
    ex = np.linalg.norm(w["x"] - l["x"]) / np.linalg.norm(l["x"])
    assert ex < (1e-10 if wide64 else 5e-3), ex


@pytest.mark.parametrize("dtype,kind,nrhs", [
    (np.float64, "poisson1000", 1),              # config 2 at full size (3907 d.q partials: the folded alpha reaches up to 4096 on these handles)
    (np.complex128, "helm500", 1),               # config 3 at full size, the build's wide type
    (np.complex64, "helm500", 1),                # ... and the reference's
    (np.float32, "poisson300", 1),
    (np.float64, "poisson300", 3),               # three right-hand sides, claimed in turn by one group
    (np.float64, "band50000", 1),                # irregular rows, last member partial
])
def test_bits_do_not_depend_on_the_call_lengths(pkg, gpu, dtype, kind, nrhs):
    """iterate(30) (chip-wide resident loop), iterate(15) twice (launched loops: calls below resident_wide_min), 16 + 14 (resident,
    then launched on its state), 14 + 16 (the other way round) and the launched loops alone (resident_wide_min out of reach) return the
    same x and the same residual history, bit for bit (results depended on the call length)."""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    if kind == "helm500":
        ip, ix, da = cg_numpy.helm_fe_var(500, 12.0, np.ones((499, 499)), 0.15, 500, 500)
    elif kind.startswith("band"):
        from test_gpu_resident import _banded_spd
        ip, ix, da = _banded_spd(int(kind[4:]), 3, 0.8, 50)
    else:
        ip, ix, da = cg_numpy.poisson2d(int(kind[7:]))
    n = len(ip) - 1
    wide_t = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    rng = np.random.default_rng(n + nrhs)
    b = np.concatenate([(1.0 + r + rand_vec(rng, n, wide_t)) for r in range(nrhs)]).astype(dtype)
    A = da.astype(dtype)

    def run(calls, wide_min=16):
        pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", wide_min))
        try:
            s = pkg.Solver(ctx, n, len(ix), A, ip, ix, nrhs)
            s.set_rhs(b, None)
            kind_ = lib.cgamd_solver_loop_launches(s.handle)
            for c in calls:
                s.iterate(c)
            out = (s.x(), s.history(), kind_)
            s.close()
            return out
        finally:
            pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))

    whole = run([30])
    assert whole[2] == 1
    for calls, wm in (([15, 15], 16), ([16, 14], 16), ([14, 16], 16), ([30], 1 << 20), ([7, 1, 22], 16)):
        other = run(calls, wm)
        assert other[2] == 1                     # the same kind of handle: the resident loop can take it over
        assert np.array_equal(whole[1], other[1]), (calls, wm, "history")
        assert np.array_equal(whole[0], other[0]), (calls, wm, "x")
    assert np.all(np.isfinite(whole[1]))


@pytest.mark.parametrize("N,nrhs", [(128, 1), (128, 9), (100, 2)])
def test_small_complex128_fe_systems_run_the_chip_wide_groups(pkg, gpu, N, nrhs):
    """helmFE_var(128) in complex128 (16 384 rows x 7 entries: 1024 rows x 20 bytes per entry do not fit the one-XCD loop's LDS) ran
    two launches per iteration until round 3.  The chip-wide groups take such handles over now: same bits as the
    handle's launched loops, oracle tolerance, one launch per call."""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    n = len(ip) - 1
    rng = np.random.default_rng(N + nrhs)
    b = np.concatenate([(1.0 + r + rand_vec(rng, n, np.complex128)) for r in range(nrhs)])

    def run(calls, wide_min=16):
        pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", wide_min))
        try:
            s = pkg.Solver(ctx, n, len(ix), da.astype(np.complex128), ip, ix, nrhs)
            s.set_rhs(b, None)
            kind = lib.cgamd_solver_loop_launches(s.handle)
            for c in calls:
                s.iterate(c)
            out = (s.x(), s.history(), kind)
            s.close()
            return out
        finally:
            pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))

    x, h, kind = run([40])
    assert kind == 1
    for calls, wm in (([20, 20], 16), ([40], 1 << 20), ([10, 30], 16)):
        x2, h2, _ = run(calls, wm)
        assert np.array_equal(h, h2) and np.array_equal(x, x2), (calls, wm)
    xo, ho = cg_oracle.cg(ip, ix, da, b, nrhs=nrhs, n_iterations=40, mode=cg_oracle.MODE_SEQUENTIAL)
    live = np.abs(ho) > 1e-6 * np.abs(ho[0])
    assert np.max((np.abs(h - ho) / np.abs(ho))[live]) < 1e-10


@pytest.mark.parametrize("n_side,nrhs,want_kind", [(128, 3, 0), (300, 1, 1)])
def test_resident_launch_that_cannot_form_its_group_falls_back_untouched(pkg, gpu, n_side, nrhs, want_kind):
    """CUs held by other work: a resident launch whose group never fills gives up before touching anything (bounded wait), and
    the handle continues with the launched loops -- same result as if the resident loop had been switched off.  Provoked with the
    test hook that launches one work-group too few."""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    ip, ix, da = cg_numpy.poisson2d(n_side)
    n = n_side * n_side
    b = np.tile(np.linspace(1.0, 2.0, n), nrhs)

    def run(knobs):
        for k, v in knobs.items():
            pkg._lib.check(lib.cgamd_tune(k.encode(), v))
        try:
            s = pkg.Solver(ctx, n, len(ix), da, ip, ix, nrhs)
            s.set_rhs(b, None)
            before = lib.cgamd_solver_loop_launches(s.handle)
            s.iterate(20)
            s.iterate(20)
            out = (s.x(), s.history(), before, lib.cgamd_solver_loop_launches(s.handle))
            s.close()
            return out
        finally:
            for k in knobs:
                pkg._lib.check(lib.cgamd_tune(k.encode(), {"resident": 1, "resident_claim_ms": 200, "resident_wide_min": 16}.get(k, 0)))

    # the launched loops of the same kind of handle (one the chip-wide loop can take over sums its partials in the members' order)
    x0, h0, _, k0 = run({"resident": 0} if want_kind == 0 else {"resident_wide_min": 1 << 20})
    x1, h1, before, after = run({"dev.resident_test_short_grid": 1, "resident_claim_ms": 40})
    assert before == want_kind and after >= 2                # resident loop chosen at first, launched loops after the failed launch
    assert k0 == (after if want_kind == 0 else 1)
    assert np.array_equal(h1, h0) and np.array_equal(x1, x0)


@pytest.mark.parametrize("dtype,kind,nrhs,calls", [
    (np.float64, "poisson300", 9, [20, 4, 17]),      # 22 work-groups per right-hand side, 9 groups at once; a launched call in between
    (np.complex64, "helm200", 9, [24]),              # 40 000 rows x 9 (a larger as_prec shape): 20 work-groups each
    (np.complex64, "helm500", 2, [20]),              # config 3 x 2: two groups of 123 work-groups, one poller per XCD and group
    (np.float64, "poisson300", 20, [18]),            # more right-hand sides than groups (11): two solves per group
])
def test_wide_resident_loop_with_several_right_hand_sides(pkg, gpu, dtype, kind, nrhs, calls):
    """one chip-wide group per right-hand side, several at once; every right-hand side against the oracle and the launched loops"""
    ctx, queue, kernels = gpu
    if kind.startswith("helm"):
        N = int(kind[4:])
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    else:
        ip, ix, da = cg_numpy.poisson2d(int(kind[7:]))
    n = len(ip) - 1
    wide_t = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    rng = np.random.default_rng(n + nrhs)
    b = np.concatenate([(r + 1) * 0.5 + rand_vec(rng, n, wide_t) for r in range(nrhs)]).astype(dtype)
    A = da.astype(dtype)
    w = _run(pkg, ctx, ip, ix, A, b, calls, True, nrhs)
    l = _run(pkg, ctx, ip, ix, A, b, calls, False, nrhs)
    assert w["kind"] == 1 and l["kind"] >= 2
    iters = sum(calls)
    cg_oracle.set_threads(16)
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide_t), b.astype(wide_t), nrhs=nrhs, n_iterations=iters, mode=cg_oracle.MODE_FAST)
    f64 = np.dtype(dtype) == np.float64
    upto = ho.shape[0] if f64 else 13
    assert w["h"].shape == ho.shape
    assert np.max((np.abs(w["h"] - ho) / np.abs(ho))[:upto]) < (1e-10 if f64 else 2e-3)
    assert np.max((np.abs(w["h"] - l["h"]) / np.abs(l["h"]))[:upto]) < (1e-11 if f64 else 2e-3)
    for r in range(nrhs):
        sl = slice(r * n, (r + 1) * n)
        ex = np.linalg.norm(w["x"][sl] - l["x"][sl]) / np.linalg.norm(l["x"][sl])
        assert ex < (1e-10 if f64 else 5e-3), (r, ex)


@pytest.mark.parametrize("dtype,kind,nrhs", [
    (np.complex64, "helm128", 1),                # the reference's sub-domain system with its Jacobi preconditioner (helmFE_var.PCG)
    (np.complex128, "helm128", 1),
    (np.float64, "poisson300", 1),
    (np.float64, "poisson300", 3),               # one group per right-hand side, the same M
    (np.complex64, "helm500", 1),                # config 3's size
])
def test_resident_jacobi_pcg_is_the_launched_pcg_bit_for_bit(pkg, gpu, dtype, kind, nrhs):
    """Jacobi-preconditioned CG (reference helmFE_var.py:546-586, M = 1 / diag) in the chip-wide resident loop: rho = r.z in alpha and
    beta, p = beta p + m r, r.r recorded.  Same x and history as the four-launch PCG loop of the same handle, however the solve is
    cut into calls (30 | 15 + 15 | 16 + 14 | launched only), the oracle restatement (bit-identical to the reference's PCG on CPU) to
    1e-10 in fp64, and the tolerance stop on the device returns the reference's (x, i)."""
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    if kind.startswith("helm"):
        N = int(kind[4:])
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    else:
        ip, ix, da = cg_numpy.poisson2d(int(kind[7:]))
        da = da.copy()
        da[ix == np.repeat(np.arange(len(ip) - 1), np.diff(ip))] += np.linspace(0.5, 4.0, len(ip) - 1)      # a diagonal worth preconditioning
    n = len(ip) - 1
    import scipy.sparse as sp
    m = 1.0 / sp.csr_matrix((da, ix, ip), shape=(n, n)).diagonal()
    wide_t = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    rng = np.random.default_rng(n + nrhs)
    B = np.concatenate([(1.0 + r + rand_vec(rng, n, wide_t)) for r in range(nrhs)])

    def run(calls, wide_min=16):
        pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", wide_min))
        try:
            s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, nrhs)
            s.set_preconditioner(m.astype(dtype))
            s.set_rhs(B.astype(dtype), None)
            kind_ = lib.cgamd_solver_loop_launches(s.handle)
            for c in calls:
                s.iterate(c)
            out = (s.x(), s.history(), kind_)
            s.close()
            return out
        finally:
            pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))

    whole = run([30])
    assert whole[2] == 1
    for calls, wm in (([15, 15], 16), ([16, 14], 16), ([14, 16], 16), ([30], 1 << 20)):
        other = run(calls, wm)
        assert np.array_equal(whole[1], other[1]), (calls, wm, "history")
        assert np.array_equal(whole[0], other[0]), (calls, wm, "x")
    wide64 = np.dtype(dtype) in (np.dtype(np.float64), np.dtype(np.complex128))
    for r in range(nrhs):
        xo, _, ho = cg_numpy.pcg_diag(ip, ix, da, B[r * n:(r + 1) * n], m, tol=0.0, maxit=30, history=True)
        keep = np.abs(ho) / np.abs(ho[0]) > (1e-8 if wide64 else 1e-3)
        assert np.max(np.abs(whole[1][keep, r] - ho[keep]) / np.abs(ho[keep])) < (1e-10 if wide64 else 2e-3), r
    if nrhs == 1 and wide64:
        # the stopping test on the device: the reference's (x, i)
        s = pkg.Solver(ctx, n, len(ix), da.astype(dtype), ip, ix, 1)
        x, i = s.pcg(B.astype(dtype), M=m.astype(dtype), tol=1e-3, maxit=400)
        s.close()
        xo, io, _ = cg_numpy.pcg_diag(ip, ix, da, B, m, tol=1e-3, maxit=400, history=True)
        assert i == io and np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-9
