"""GPU tests of the row-partitioned C loop (csrc/dist.cpp) that one GPU can run: world size 1 with and
without an RCCL communicator (all-reduce of one rank, no neighbours), against the single-GPU solver and the
oracle; and the generic cg_loop driven by the HIP kernels (HipOps) on one rank.  Ranks > 1 need more GPUs
than the test box has: that path is covered on CPU by tests/test_dist_gloo.py (same plan, same recurrence)."""
import importlib

import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import PKG_NAME

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("dtype", [np.float64, np.complex128, np.float32])
def test_dist_loop_world1_matches_oracle(pkg, gpu, dtype, with_comm):
    import torch
    ctx, queue, kernels = gpu
    dmod = importlib.import_module(PKG_NAME + ".dist")
    lib = pkg._lib.load()
    if np.dtype(dtype).kind == "c":
        N = 20
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = cg_numpy.rhsA(N, 12.0).flatten()
    else:
        ip, ix, da = cg_numpy.laplace3d(9, 8, 7)
        b = np.linspace(1.0, 2.0, len(ip) - 1)
    n = len(ip) - 1
    dev = torch.device("cuda", 0)
    plan = dmod.build_halo_plan(torch.from_numpy(ix.astype(np.int64)).to(dev), [(0, n)], 0)
    uid = None
    if with_comm:
        buf = np.zeros(128, dtype=np.uint8)
        pkg._lib.check(lib.cgamd_comm_unique_id(pkg._lib.ptr(buf)))
        uid = buf
    tdt = pkg.generators.torch_dtype(dtype)
    vals = torch.from_numpy(da.astype(dtype)).to(dev)
    indptr = torch.from_numpy(ip).to(dev)
    s = dmod.DistSolver(ctx, plan, indptr, vals, dtype, unique_id=uid, flags=pkg._lib.DIST_GRAPH if with_comm else 0)
    bl = torch.from_numpy(b.astype(dtype)).to(dev)
    iters = 20
    s.set_rhs(bl, None)
    s.iterate(iters)
    x = s.x(torch.empty(n, dtype=tdt, device=dev)).cpu().numpy()
    h = s.history()
    s.close()
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    tol = 1e-4 if np.dtype(dtype) == np.float32 else 1e-10
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > 1e-4
    assert np.max(np.abs(h[keep] - ho[keep, 0]) / np.abs(ho[keep, 0])) < tol
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < (1e-3 if tol > 1e-6 else 1e-9)


def test_generic_loop_with_hip_ops_world1(pkg, gpu):
    import torch
    ctx, queue, kernels = gpu
    dmod = importlib.import_module(PKG_NAME + ".dist")
    ip, ix, da = cg_numpy.poisson2d(30)
    n = 900
    b = np.linspace(1.0, 2.0, n)
    dev = torch.device("cuda", 0)
    plan = dmod.build_halo_plan(torch.from_numpy(ix.astype(np.int64)).to(dev), [(0, n)], 0)
    ops = dmod.HipOps(ctx, plan, torch.from_numpy(ip).to(dev), torch.from_numpy(da).to(dev), np.float64)
    bl = torch.from_numpy(b).to(dev)
    x, hist = dmod.cg_loop(ops, dmod.TorchComm(plan), plan, bl, torch.zeros_like(bl), 15)
    xo, ho = cg_oracle.cg(ip, ix, da, b, n_iterations=15, mode=cg_oracle.MODE_SEQUENTIAL)
    assert np.max(np.abs(hist.cpu().numpy() - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-10
    assert np.linalg.norm(x.cpu().numpy() - xo) / np.linalg.norm(xo) < 1e-9


def test_device_generators_match_oracle_generators(pkg, gpu):
    ctx, queue, kernels = gpu
    for dtype in (np.float64, np.complex64):
        ipd, ixd, dad = pkg.generators.laplace3d(ctx, 7, 5, 6, dtype=dtype)
        ip, ix, da = cg_numpy.laplace3d(7, 5, 6, dtype=dtype)
        assert np.array_equal(ipd.cpu().numpy(), ip) and np.array_equal(ixd.cpu().numpy(), ix)
        assert np.array_equal(dad.cpu().numpy(), da)
    # a slab with global column ids (the multi-GPU partition input)
    ipd, ixd, dad = pkg.generators.laplace3d(ctx, 7, 5, 6, row_begin=50, row_end=140)
    ip, ix, da = cg_numpy.laplace3d(7, 5, 6)
    assert np.array_equal(ipd.cpu().numpy(), ip[50:141] - ip[50])
    assert np.array_equal(ixd.cpu().numpy(), ix[ip[50]:ip[140]])
    ipd, ixd, dad = pkg.generators.poisson2d(ctx, 13)
    ip, ix, da = cg_numpy.poisson2d(13)
    assert np.array_equal(ipd.cpu().numpy(), ip) and np.array_equal(ixd.cpu().numpy(), ix)
    assert np.array_equal(dad.cpu().numpy(), da)


@pytest.mark.parametrize("flags", ["overlap", "no_overlap", "graph"])
def test_exchange_and_overlap_through_self_halo(pkg, gpu, flags):
    """One GPU cannot host two RCCL ranks, but a rank may be its own peer: the columns < h referenced by rows >= h
    are routed through halo slots that the rank fills from its own first h entries with ncclSend/ncclRecv to
    self.  The product is unchanged, so the run must reproduce the plain single-GPU solver -- while exercising
    pack, grouped send/recv, the interior/boundary row-block split on two streams (plain launches) and the
    hipGraph replay of the in-line exchange."""
    import torch
    ctx, queue, kernels = gpu
    dmod = importlib.import_module(PKG_NAME + ".dist")
    lib = pkg._lib.load()
    nx, ny, nz = 20, 20, 30
    ip, ix, da = cg_numpy.laplace3d(nx, ny, nz)
    n, h = nx * ny * nz, nx * ny
    rows = np.repeat(np.arange(n), np.diff(ip))
    route = (ix < h) & (rows >= h)
    assert route.sum() == h                      # exactly the plane below
    cols_local = np.where(route, n + ix, ix).astype(np.int32)
    dev = torch.device("cuda", 0)
    plan = dmod.HaloPlan(0, 1, 0, n, n, h, torch.from_numpy(cols_local).to(dev), torch.arange(h), [0], [h], [h],
                         torch.arange(h, dtype=torch.int32, device=dev))
    buf = np.zeros(128, dtype=np.uint8)
    pkg._lib.check(lib.cgamd_comm_unique_id(pkg._lib.ptr(buf)))
    fl = {"overlap": 0, "no_overlap": pkg._lib.DIST_NO_OVERLAP, "graph": pkg._lib.DIST_GRAPH}[flags]
    vals, indptr = torch.from_numpy(da).to(dev), torch.from_numpy(ip).to(dev)
    s = dmod.DistSolver(ctx, plan, indptr, vals, np.float64, unique_id=buf, flags=fl)
    b = np.linspace(1.0, 2.0, n)
    bl = torch.from_numpy(b).to(dev)
    iters = 30
    s.set_rhs(bl, None)
    s.iterate(iters)
    x = s.x(torch.empty(n, dtype=torch.float64, device=dev)).cpu().numpy()
    hist = s.history()
    s.close()
    xo, ho = cg_oracle.cg(ip, ix, da, b, n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    assert np.max(np.abs(hist - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-10
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-9
