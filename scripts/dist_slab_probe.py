#!/usr/bin/env python3
"""What one rank of an N-GPU strong-scaling run costs per iteration, measured on ONE GPU.

The rank's z-slab of the 250x200x200 system (nz/N planes) is solved (a) by the plain single-GPU solver (no
communication: the floor) and (b) by every distributed loop with the rank as its own halo peer (the plane below is
routed through halo slots the rank fills from its own first plane), so pack/push/wait/unpack, the
interior/boundary split and the scalar all-reduce kernels all run -- only the xGMI wire latency is missing.
usage: python scripts/dist_slab_probe.py [--grid 250x200x25] [--iters 400]
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="250x200x25")
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--skip-rccl", action="store_true")
    ap.add_argument("--only", default="", help="run just this loop (e.g. 'p2p+graph'); skips the plain solver too")
    ap.add_argument("--skip-staged", action="store_true", help="skip the staged peer-to-peer loops (seven launches)")
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
    dmod = importlib.import_module("conjugate-gradient-pyopencl_amd.dist")
    L = pkg._lib
    lib = L.load()
    for kv in filter(None, os.environ.get("CG_TUNE", "").split(",")):      # CG_TUNE=key=value,... like bench.py and the tests
        k, v = kv.split("=")
        L.check(lib.cgamd_tune(k.encode(), int(v)))
    nx, ny, nz = (int(v) for v in args.grid.split("x"))
    n, h = nx * ny * nz, nx * ny
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    indptr, indices, data = pkg.generators.laplace3d(ctx, nx, ny, nz, dtype=np.float64)
    nnz = int(indices.numel())
    b = torch.full((n,), 5.0, dtype=torch.float64, device=dev)

    def timed(obj):
        best = []
        for _ in range(args.rounds):
            obj.set_rhs(b, None) if not isinstance(obj, pkg.Solver) else obj.set_rhs(b, None, on_device=True)
            obj.iterate(20)
            ctx.synchronize() if isinstance(obj, pkg.Solver) else obj.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            obj.iterate(args.iters)
            ctx.synchronize() if isinstance(obj, pkg.Solver) else obj.synchronize()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) / args.iters * 1e6)
        return min(best), float(np.median(best))

    for name, fl in (("hipGraph", 0), ("plain launches", L.NO_GRAPH)):
        if args.only and args.only != "plain " + name:
            continue
        s = pkg.Solver(ctx, n, nnz, data, indptr, indices, 1, flags=L.MATRIX_ON_DEVICE | fl, dtype=np.float64)
        lo, med = timed(s)
        print(f"{args.grid} rows={n}  plain solver ({name:14s})        : {lo:7.2f} us/iter (median {med:7.2f})", flush=True)
        s.close()

    # the slab loop (one launch per call, vectors in registers) on the rank's matrix without the self-halo: what the rank's own work costs
    if not args.only or args.only == "slab":
        plan0 = dmod.build_halo_plan(indices.long(), [(0, n)], 0)
        d = dmod.DistSolver(ctx, plan0, indptr, data, np.float64, flags=L.DIST_RESIDENT)
        lo, med = timed(d)
        print(f"{args.grid} rows={n}  slab loop (no halo), launches/iteration {lib.cgamd_dist_loop_launches(d.handle)}: {lo:7.2f} us/iter (median {med:7.2f})", flush=True)
        d.close()

    # self-halo routing: columns < h referenced by rows >= h go through halo slots
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (indptr[1:] - indptr[:-1]).long())
    route = (indices < h) & (rows >= h)
    cols_local = torch.where(route, indices + n, indices).to(torch.int32)
    plan = dmod.HaloPlan(0, 1, 0, n, n, h, cols_local, torch.arange(h), [0], [h], [h],
                         torch.arange(h, dtype=torch.int32, device=dev))
    ST = L.DIST_P2P_STAGED
    SR = L.DIST_SINGLE_REDUCTION
    modes = [("p2p slab", "p2p", L.DIST_RESIDENT),
             ("p2p2 single-reduction", "p2p", SR), ("p2p2 single-red.+graph", "p2p", SR | L.DIST_GRAPH),
             ("p2p4", "p2p", 0), ("p2p4+graph", "p2p", L.DIST_GRAPH),
             ("p2p staged", "p2p", ST | L.DIST_NO_OVERLAP), ("p2p staged+graph", "p2p", ST | L.DIST_GRAPH | L.DIST_NO_OVERLAP),
             ("p2p staged overlap", "p2p", ST)]
    if not args.skip_rccl:
        modes += [("rccl single-red.+graph", "rccl", SR | L.DIST_GRAPH), ("rccl single-reduction", "rccl", SR),
                  ("rccl+graph", "rccl", L.DIST_GRAPH), ("rccl", "rccl", 0)]
    for name, comm, fl in modes:
        if args.only and args.only != name:
            continue
        if args.skip_staged and "staged" in name:
            continue
        uid = np.zeros(128, dtype=np.uint8)
        if comm == "rccl":
            L.check(lib.cgamd_comm_unique_id(L.ptr(uid)))       # one communicator per solver
        d = dmod.DistSolver(ctx, plan, indptr, data, np.float64, unique_id=uid if comm == "rccl" else None, flags=fl, comm=comm)
        lo, med = timed(d)
        print(f"{args.grid} rows={n}  dist loop {name:22s}: {lo:7.2f} us/iter (median {med:7.2f})  [{lib.cgamd_dist_loop_launches(d.handle)} launches/iteration]", flush=True)
        d.close()


if __name__ == "__main__":
    main()
