"""Slab loop (csrc/slab.hip, CGAMD_DIST_RESIDENT): every iteration of an iterate() call in ONE launch with the vectors in
registers and the matrix streamed -- the loop body of the reference (clcg.c:297-419) for systems of ~1M to ~3M rows (one rank's
slab of the row-partitioned headline system).  Same recurrence and per-row order as the launched loops; partial sums are per
member, so results are held to the oracle at the standard tolerance (fp64 delta_k rtol 1e-10, x 1e-9), are bitwise
reproducible run to run, and a handle may alternate between this loop and the launched one."""
import importlib

import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import PKG_NAME

pytestmark = pytest.mark.gpu
RESIDENT = 512


def _handle(pkg, ctx, ip, ix, da, dtype, flags):
    import torch
    dmod = importlib.import_module(PKG_NAME + ".dist")
    dev = torch.device("cuda", 0)
    n = len(ip) - 1
    plan = dmod.build_halo_plan(torch.from_numpy(ix.astype(np.int64)).to(dev), [(0, n)], 0)
    vals = torch.from_numpy(da.astype(dtype)).to(dev)
    indptr = torch.from_numpy(ip.astype(np.int32)).to(dev)
    s = dmod.DistSolver(ctx, plan, indptr, vals, dtype, flags=flags)
    s._test_keep = (vals, indptr, plan)
    return s


def _tuned(pkg, **kv):
    lib = pkg._lib.load()
    for k, v in kv.items():
        pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))


@pytest.mark.parametrize("dtype,coded,vcoded", [(np.float64, True, True), (np.float32, True, True), (np.complex64, True, True),
                                                 (np.float64, True, False), (np.float32, True, False)])
def test_slab_loop_matches_oracle(pkg, gpu, dtype, coded, vcoded):
    """vcoded: the members keep one-byte VALUE codes of their rows in LDS and stream nothing of the matrix (matrices of at most 256
    distinct entries, build_value_codes); False: the value slices are streamed (`dev.value_codes = 0`: what a variable-coefficient
    matrix gets)."""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    dev = torch.device("cuda", 0)
    if np.dtype(dtype).kind == "c":
        N = 1000                # helmFE_var(1000): 1M rows, 7 entries per row, complex symmetric
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = np.tile(cg_numpy.rhsA(100, 12.0).flatten(), 100)
    else:
        ip, ix, da = cg_numpy.laplace3d(120, 100, 90)     # 1.08M rows: z-neighbours 12 000 rows away, other members' columns
        b = np.linspace(1.0, 2.0, len(ip) - 1)
    n = len(ip) - 1
    tdt = pkg.generators.torch_dtype(dtype)
    try:
        if not coded:
            _tuned(pkg, index_codes=0)
        _tuned(pkg, **{"dev.value_codes": int(vcoded)})
        s = _handle(pkg, ctx, ip, ix, da, dtype, RESIDENT)
    finally:
        _tuned(pkg, index_codes=1, **{"dev.value_codes": 1})
    assert (s.index_codes() > 0) == coded
    assert lib.cgamd_dist_loop_launches(s.handle) == 0
    bl = torch.from_numpy(b.astype(dtype)).to(dev)
    outs = []
    for split in ((20, 16), (36,)):
        s.set_rhs(bl, None)
        for k in split:
            s.iterate(k)
        outs.append((s.x(torch.empty(n, dtype=tdt, device=dev)).cpu().numpy(), s.history().copy()))
    assert lib.cgamd_dist_loop_launches(s.handle) == 0          # no launch fell back
    # short calls (below resident_wide_min) take the launched loop on the same handle and state
    s.set_rhs(bl, None)
    s.iterate(20)
    s.iterate(5)
    s.iterate(11)
    mixed = (s.x(torch.empty(n, dtype=tdt, device=dev)).cpu().numpy(), s.history().copy())
    s.close()
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=36, mode=cg_oracle.MODE_FAST)
    single = np.dtype(dtype) in (np.dtype(np.float32), np.dtype(np.complex64))
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > (1e-4 if single else 1e-8)
    if np.dtype(dtype) == np.dtype(np.complex64):
        # complex64 on this indefinite system: the residual GROWS (delta_36 = 15 delta_0) and single-precision rounding with it; measured
        # on the device (scripts/dev/slab_c64_diag.py): slab loop and launched loop drift from the fp64 oracle identically (1.35e-4 at
        # k = 24, 3.5e-4 at k = 36) and equal each other bit for bit (fp64 accumulators rounded to complex64).  Held: the launched
        # loop of the same handle type to 1e-6, the oracle to 1e-3
        ref = _handle(pkg, ctx, ip, ix, da, dtype, 0)
        ref.set_rhs(bl, None)
        ref.iterate(36)
        href = ref.history().copy()
        ref.close()
        for _, h in outs + [mixed]:
            assert np.max(np.abs(h - href) / np.abs(href)) < 1e-6
            assert np.max(np.abs(h - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-3
    else:
        for x, h in outs + [mixed]:
            assert np.max(np.abs(h[keep] - ho[keep, 0]) / np.abs(ho[keep, 0])) < (1e-4 if single else 1e-10)
            assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < (1e-3 if single else 1e-9)
    # call lengths do not change the bits (the state handed from one launch to the next is the whole state)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_slab_loop_is_reproducible_and_flag_is_inert_where_it_does_not_apply(pkg, gpu):
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    dev = torch.device("cuda", 0)
    ip, ix, da = cg_numpy.laplace3d(120, 100, 90)
    n = len(ip) - 1
    bl = torch.from_numpy(np.linspace(1.0, 2.0, n)).to(dev)
    runs = []
    for _ in range(2):
        s = _handle(pkg, ctx, ip, ix, da, np.float64, RESIDENT)
        s.set_rhs(bl, None)
        s.iterate(40)
        runs.append((s.x(torch.empty(n, dtype=torch.float64, device=dev)).cpu().numpy(), s.history().copy()))
        s.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    # a small system (one member would do): the flag changes nothing, the launched loop runs
    ip, ix, da = cg_numpy.laplace3d(9, 8, 7)
    s = _handle(pkg, ctx, ip, ix, da, np.float64, RESIDENT)
    assert lib.cgamd_dist_loop_launches(s.handle) > 0
    s.close()


@pytest.mark.parametrize("dtype", [np.float64, np.complex64])
def test_slab_loop_with_the_exchange_through_self_halo(pkg, gpu, dtype):
    """the rank as its own halo peer over the peer-to-peer mailboxes: boundary pushes into the tail of the published-d buffers,
    halo flags, and the rank-ordered scalar exchange of both reductions all run for real inside the one launch"""
    import torch
    ctx, queue, kernels = gpu
    lib = pkg._lib.load()
    dmod = importlib.import_module(PKG_NAME + ".dist")
    dev = torch.device("cuda", 0)
    if np.dtype(dtype).kind == "c":
        N = 1000
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = np.tile(cg_numpy.rhsA(100, 12.0).flatten(), 100)
        h = N + 1
    else:
        nx, ny, nz = 120, 100, 90
        ip, ix, da = cg_numpy.laplace3d(nx, ny, nz)
        b = np.linspace(1.0, 2.0, len(ip) - 1)
        h = nx * ny
    n = len(ip) - 1
    rows = np.repeat(np.arange(n), np.diff(ip))
    route = (ix < h) & (rows >= h)
    cols_local = np.where(route, n + ix, ix).astype(np.int32)
    plan = dmod.HaloPlan(0, 1, 0, n, n, h, torch.from_numpy(cols_local).to(dev), torch.arange(h), [0], [h], [h],
                         torch.arange(h, dtype=torch.int32, device=dev))
    vals = torch.from_numpy(da.astype(dtype)).to(dev)
    indptr = torch.from_numpy(ip.astype(np.int32)).to(dev)
    s = dmod.DistSolver(ctx, plan, indptr, vals, dtype, flags=RESIDENT, comm="p2p")
    assert lib.cgamd_dist_loop_launches(s.handle) == 0
    tdt = pkg.generators.torch_dtype(dtype)
    bl = torch.from_numpy(b.astype(dtype)).to(dev)
    iters = 24
    outs = []
    for split in ((24,), (16, 4, 4)):       # 4 < resident_wide_min: the launched four-launch loop continues on the same state and epochs
        s.set_rhs(bl, None)
        for k in split:
            s.iterate(k)
        outs.append((s.x(torch.empty(n, dtype=tdt, device=dev)).cpu().numpy(), s.history().copy()))
        assert s.p2p_error() == 0
    s.close()
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=iters, mode=cg_oracle.MODE_FAST)
    single = np.dtype(dtype) == np.dtype(np.complex64)
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > (1e-4 if single else 1e-8)
    for x, hist in outs:
        # (complex64 on this system drifts from the fp64 oracle by 1.35e-4 at k = 24 in every loop: see test_slab_loop_matches_oracle)
        assert np.max(np.abs(hist[keep] - ho[keep, 0]) / np.abs(ho[keep, 0])) < (1e-3 if single else 1e-10)
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < (1e-3 if single else 1e-9)
    assert np.array_equal(outs[0][1][:17], outs[1][1][:17])         # the first 16 iterations ran in the slab loop both times
