"""CPU tests of the multi-GPU host logic with world_size 2 and 3 over gloo: row ranges, halo plan
(send/recv lists, column renumbering) and the distributed recurrence with its boundary exchange and
scalar all-reduces.  The local kernels are the CPU oracle's (tests may use the oracle); the comm layer,
the plan and the loop are the product's (conjugate-gradient-pyopencl_amd/dist.py), the same code the
GPU path feeds to the RCCL loop in csrc/dist.cpp."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import PKG_NAME, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _system(kind):
    import cg_numpy
    if kind == "lap3d":
        ip, ix, da = cg_numpy.laplace3d(6, 5, 9)
        b = np.linspace(1.0, 2.0, len(ip) - 1)
    elif kind == "helm":
        N = 12
        ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
        b = cg_numpy.rhsA(N, 12.0).flatten()
    else:   # random sparse, unsorted columns, empty rows: irregular halos with several peers
        from conftest import rand_csr
        rng = np.random.default_rng(11)
        ip, ix, da = rand_csr(rng, 157, 4, np.float64, empty_rows=True)
        import scipy.sparse as sp
        A = sp.csr_matrix((da, ix, ip), shape=(157, 157))
        A = sp.csr_matrix(A + A.T + sp.identity(157) * 40.0)
        ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
        b = rng.standard_normal(157)
    return ip, ix, da, b


def _worker(rank, world, port, kind, iters, out_dir, single_reduction=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import cg_oracle
    pkg = importlib.import_module(PKG_NAME)
    dmod = importlib.import_module(PKG_NAME + ".dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ip, ix, da, b = _system(kind)
        n = len(ip) - 1
        ranges = dmod.row_ranges(n, world, indptr=ip if kind == "random" else None)
        rb, re = ranges[rank]
        lo, hi = ip[rb], ip[re]
        ip_loc = (ip[rb:re + 1] - lo).astype(np.int32)
        plan = dmod.build_halo_plan(torch.from_numpy(ix[lo:hi].astype(np.int64)), ranges, rank)
        cols_loc = plan.cols_local.numpy()
        vals_loc = da[lo:hi]
        # ---- plan invariants
        assert plan.n_local == re - rb and sum(plan.recv_counts) == plan.n_halo
        assert cols_loc.min(initial=0) >= 0 and cols_loc.max(initial=0) < plan.n_local + plan.n_halo
        glob = np.where(cols_loc < plan.n_local, cols_loc + rb, plan.halo_global.numpy()[np.maximum(cols_loc - plan.n_local, 0)])
        assert np.array_equal(glob, ix[lo:hi])
        assert len(plan.send_index) == sum(plan.send_counts)
        assert rank not in plan.peers and plan.peers == sorted(plan.peers)
        allp = [None] * world
        dist.all_gather_object(allp, dict(zip(plan.peers, zip(plan.send_counts, plan.recv_counts))))
        for p, (sc, rc) in allp[rank].items():          # what I send to p is what p receives from me
            assert allp[p][rank] == (rc, sc)

        class OracleOps:
            def spmv(self, v_ext):
                ne = plan.n_local + plan.n_halo
                ptr_ext = np.concatenate([ip_loc, np.full(plan.n_halo, ip_loc[-1], dtype=np.int32)])
                y = cg_oracle.spmv(ptr_ext, cols_loc, vals_loc, v_ext.numpy(), mode=cg_oracle.MODE_SEQUENTIAL)
                return torch.from_numpy(y[:plan.n_local])

            def dot(self, a, c):
                return torch.from_numpy(cg_oracle.vdot(a.numpy().copy(), c.numpy().copy(), mode=cg_oracle.MODE_SEQUENTIAL))[0]

        comm = dmod.TorchComm(plan)
        bl = torch.from_numpy(b[rb:re].astype(da.dtype))
        loop = dmod.cg_loop_single_reduction if single_reduction else dmod.cg_loop
        x, hist = loop(OracleOps(), comm, plan, bl, torch.zeros_like(bl), iters)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), x=x.numpy(), hist=hist.numpy(), rb=rb, re=re,
                 n_halo=plan.n_halo, peers=np.array(plan.peers))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "lap3d"), (3, "lap3d"), (2, "helm"), (3, "random")])
def test_distributed_cg_matches_serial_oracle(tmp_path, world, kind):
    import torch.multiprocessing as mp
    import cg_oracle
    iters = 12
    mp.spawn(_worker, args=(world, _free_port(), kind, iters, str(tmp_path)), nprocs=world, join=True)
    ip, ix, da, b = _system(kind)
    xo, ho = cg_oracle.cg(ip, ix, da, b.astype(da.dtype), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    x = np.concatenate([p["x"] for p in parts])
    if kind != "random":
        assert [int(p["rb"]) for p in parts] == [len(b) * g // world for g in range(world)]
    # every rank holds the same global residual history
    for p in parts[1:]:
        assert np.allclose(p["hist"], parts[0]["hist"], rtol=1e-12)
    # fp64 tolerance of the north star: delta_k rtol 1e-10 (reduction order differs: per-rank partial sums)
    assert np.max(np.abs(parts[0]["hist"] - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-10
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-9
    if kind == "lap3d":       # z-slab partition of a 7-point stencil: one plane per neighbour
        assert all(int(p["n_halo"]) in (30, 60) for p in parts)
        assert [list(p["peers"]) for p in parts] == ([[1], [0]] if world == 2 else [[1], [0, 2], [1]])


@pytest.mark.parametrize("world,kind", [(2, "lap3d"), (3, "helm"), (2, "random")])
def test_single_reduction_loop_matches_serial_oracle(tmp_path, world, kind):
    """the single-reduction form of the distributed recurrence (dist.cg_loop_single_reduction, the host twin of csrc/cg1.hip):
    one all-reduce of {r.r, w.r} per iteration, the halo exchange carries r.  Same iterates as the reference recurrence to the
    stated fp64 tolerance (not bit for bit: the rounding differs)."""
    import torch.multiprocessing as mp
    import cg_oracle
    iters = 12
    mp.spawn(_worker, args=(world, _free_port(), kind, iters, str(tmp_path), True), nprocs=world, join=True)
    ip, ix, da, b = _system(kind)
    xo, ho = cg_oracle.cg(ip, ix, da, b.astype(da.dtype), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    parts = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    x = np.concatenate([p["x"] for p in parts])
    for p in parts[1:]:
        assert np.allclose(p["hist"], parts[0]["hist"], rtol=1e-12)
    assert np.max(np.abs(parts[0]["hist"] - ho[:, 0]) / np.abs(ho[:, 0])) < 1e-10
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-9


def test_nnz_balanced_row_ranges(pkg):
    """irregular matrices: contiguous row blocks with (nearly) equal non-zero counts"""
    dmod = importlib.import_module(PKG_NAME + ".dist")
    rng = np.random.default_rng(2)
    counts = np.concatenate([rng.integers(1, 4, 700), rng.integers(40, 60, 300)])        # a dense tail
    ip = np.concatenate([[0], np.cumsum(counts)])
    n, nnz = len(counts), int(ip[-1])
    for world in (2, 3, 8):
        rr = dmod.row_ranges(n, world, indptr=ip)
        assert rr[0][0] == 0 and rr[-1][1] == n and all(a[1] == b[0] for a, b in zip(rr, rr[1:]))
        per = [int(ip[e] - ip[b]) for b, e in rr]
        assert all(e > b for b, e in rr)
        assert max(per) - min(per) <= 2 * counts.max(), (world, per)
        assert max(per) < 1.1 * nnz / world + counts.max()
        rows_equal = [int(ip[e] - ip[b]) for b, e in dmod.row_ranges(n, world)]
        assert max(rows_equal) > 1.5 * max(per) or world == 2          # equal row counts would be badly unbalanced here
    # degenerate shapes: more ranks than "shares", empty rows, all weight in one row
    assert dmod.row_ranges(3, 3, indptr=[0, 0, 0, 9]) == [(0, 1), (1, 2), (2, 3)]
    assert dmod.row_ranges(4, 2, indptr=[0, 100, 100, 100, 100]) == [(0, 1), (1, 4)]
    with pytest.raises(ValueError):
        dmod.row_ranges(4, 2, indptr=[0, 1, 2])


def test_row_ranges_and_single_rank_plan(pkg):
    import torch
    dmod = importlib.import_module(PKG_NAME + ".dist")
    assert dmod.row_ranges(10, 3) == [(0, 3), (3, 6), (6, 10)]
    assert dmod.row_ranges(10_000_000, 8)[7] == (8_750_000, 10_000_000)
    cols = torch.tensor([0, 1, 1, 2, 0, 2], dtype=torch.int32)
    plan = dmod.build_halo_plan(cols, [(0, 3)], 0)
    assert plan.n_halo == 0 and plan.peers == [] and torch.equal(plan.cols_local, cols)
