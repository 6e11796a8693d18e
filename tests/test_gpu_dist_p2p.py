"""GPU tests of the row-partitioned C loop with MORE THAN ONE RANK on the one-GPU test box.  RCCL refuses two
ranks on one device, the peer-to-peer backend does not care: the ranks are separate processes that share
cuda:0, exchange halos and reduce scalars through each other's IPC mailboxes (csrc/p2p.hip "Peer-to-peer
communication").  Same plan, same loop, same kernels as the RCCL path; results against the serial oracle."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _system(kind):
    import cg_numpy
    if kind.endswith("_f32") or kind.endswith("_c64"):          # single-precision variants of the same systems
        ip, ix, da, b = _system(kind[:-4])
        dt = np.float32 if kind.endswith("_f32") else np.complex64
        return ip, ix, da.astype(dt), b.astype(dt)
    if kind == "rand":               # ~9 entries per row, random pattern: every rank is every other rank's halo peer
        import scipy.sparse as sp
        rng = np.random.default_rng(12)
        n = 7000
        P = sp.random(n, n, density=4.0 / n, random_state=rng, format="csr")
        P = P + P.T
        A = sp.csr_matrix(P + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 1.0))
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, np.linspace(1.0, 2.0, n)
    if kind == "dense":              # ~60 entries per row: the chunked row-block kernel; no halo flags -> staged in-line exchange
        import scipy.sparse as sp
        rng = np.random.default_rng(11)
        n = 5000
        P = sp.random(n, n, density=30.0 / n, random_state=rng, format="csr")
        P = P + P.T
        A = sp.csr_matrix(P + sp.diags(np.asarray(abs(P).sum(axis=1)).ravel() + 1.0))
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, np.linspace(1.0, 2.0, n)
    if kind == "lap3d":
        ip, ix, da = cg_numpy.laplace3d(24, 20, 36)          # 17280 rows: overlap lists are built (>= 16 row blocks/rank)
        b = np.linspace(1.0, 2.0, len(ip) - 1)
    else:
        N = 60
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = cg_numpy.rhsA(N, 12.0).flatten()
    return ip, ix, da, b


def _worker(rank, world, port, kind, iters, flags, out_dir, coded=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module(PKG_NAME)
    dmod = importlib.import_module(PKG_NAME + ".dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ctx = pkg.Context(0)
        ip, ix, da, b = _system(kind)
        n = len(ip) - 1
        ranges = dmod.row_ranges(n, world)
        rb, re = ranges[rank]
        lo, hi = ip[rb], ip[re]
        plan = dmod.build_halo_plan(torch.from_numpy(ix[lo:hi].astype(np.int64)), ranges, rank)
        plan.cols_local = plan.cols_local.to(dev)
        plan.send_index = plan.send_index.to(dev)
        indptr = torch.from_numpy((ip[rb:re + 1] - lo).astype(np.int32)).to(dev)
        vals = torch.from_numpy(da[lo:hi]).to(dev)
        if coded:           # one-byte column codes also for these small local matrices (default: above 32 MB)
            pkg._lib.check(pkg._lib.load().cgamd_tune(b"index_codes_min_mb", 0))
        if flags & 512:     # slab loop: the ranks' resident launches must run at the same time on the shared GPU
            pkg._lib.check(pkg._lib.load().cgamd_tune(b"dev.resident_lock", 0))
        s = dmod.DistSolver(ctx, plan, indptr, vals, da.dtype, flags=flags, comm="p2p")
        assert (s.index_codes() > 0) == coded, s.index_codes()
        if flags & 512:
            assert pkg._lib.load().cgamd_dist_loop_launches(s.handle) == 0
        bl = torch.from_numpy(b[rb:re].astype(da.dtype)).to(dev)
        s.set_rhs(bl, None)
        for part in ((16, 4, 10) if flags & 512 else (iters // 3,) * 3):      # slab loop: calls of >= 16 iterations; the 4 run launched
            s.iterate(part)
        x = s.x(torch.empty(plan.n_local, dtype=bl.dtype, device=dev)).cpu().numpy()
        hist = s.history()
        err = s.p2p_error()
        s.close()
        ctx.close()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=x, hist=hist, err=err)
        dist.barrier()
    finally:
        dist.destroy_process_group()


# flags: 0 = four-launch iteration (push + wait inside the SpMV launch, halo read in place, beta all-reduce inside aypx),
# 8 = the same from a hipGraph, 128 = staged push / unpack / all-reduce launches (+8 graph, +32 no interior/boundary overlap)
# coded: the local matrices (halo columns included) through one-byte column codes (cgamd_dist_index_codes)
@pytest.mark.parametrize("world,kind,flags,coded", [(2, "lap3d", 0, False), (3, "lap3d", 8, False), (2, "helm", 8, False), (3, "helm", 0, False),
                                                    (4, "lap3d", 0, False), (2, "lap3d", 128, False), (3, "lap3d", 128 | 8, False),
                                                    (3, "helm", 128 | 32, False), (2, "lap3d_f32", 0, False), (3, "helm_c64", 0, False),
                                                    (2, "dense", 0, False), (3, "dense", 8, False), (4, "rand", 0, False), (3, "rand", 8, False),
                                                    (3, "rand", 128, False),
                                                    (3, "lap3d", 0, True), (2, "helm", 8, True), (3, "lap3d", 128, True), (3, "helm_c64", 0, True),
                                                    (2, "lap3d", 128 | 32, True),
                                                    # 256 = single-reduction loop (csrc/cg1.hip): two launches, one scalar exchange per iteration
                                                    (2, "lap3d", 256, False), (3, "helm", 256 | 8, False), (4, "lap3d", 256, True),
                                                    (3, "helm_c64", 256, False), (2, "lap3d_f32", 256 | 8, False), (4, "rand", 256, False),
                                                    # 512 = slab loop (csrc/slab.hip): whole calls in one launch per rank; needs the column codes
                                                    (2, "lap3d", 512, True), (4, "lap3d", 512, True), (3, "helm_c64", 512, True), (3, "lap3d", 512, True),
                                                    (2, "lap3d_f32", 512, True)])
def test_p2p_multirank_on_one_gpu(tmp_path, world, kind, flags, coded):
    import torch.multiprocessing as mp
    import cg_oracle
    iters = 30
    mp.spawn(_worker, args=(world, _free_port(), kind, iters, flags, str(tmp_path), coded), nprocs=world, join=True)
    ip, ix, da, b = _system(kind)
    # single-precision runs are held against the fp64 oracle (stated tolerance 1e-4 while delta_k/delta_0 > 1e-4)
    wide = np.complex128 if np.dtype(da.dtype).kind == "c" else np.float64
    xo, ho = cg_oracle.cg(ip, ix, da.astype(wide), b.astype(wide), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    assert all(int(p["err"]) == 0 for p in parts)
    for p in parts[1:]:
        assert np.array_equal(p["hist"], parts[0]["hist"])        # rank-ordered sums: bitwise identical on all ranks
    single = np.dtype(da.dtype).itemsize <= 8 and np.dtype(da.dtype) in (np.dtype(np.float32), np.dtype(np.complex64))
    keep = np.abs(ho[:, 0]) / np.abs(ho[0, 0]) > (1e-4 if single else 1e-8)
    assert np.max(np.abs(parts[0]["hist"][keep] - ho[keep, 0]) / np.abs(ho[keep, 0])) < (1e-4 if single else 1e-10)
    x = np.concatenate([p["x"] for p in parts])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < (1e-3 if single else 1e-9)
