#!/usr/bin/env python3
"""In-process interleaved A/B of SpMV / CG-iteration variants (cdna_hip_programming.md §5.4 rule 24).
usage: python scripts/spmv_ab.py [--grid 250x200x200] [--dtype f64] [--rounds 5] cfg1 cfg2 ...
  where cfg = comma-separated key=value tuning pairs, e.g.  index_codes=0  dev.generic_spmv=1,dev.spmv_grid=1024
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
    ap.add_argument("--grid", default="250x200x200")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--helm", type=int, default=0, help="use the Helmholtz FE matrix helm_fe_var(N) instead of the 3-D stencil")
    ap.add_argument("--stencil27", type=int, default=0, help="use the 3-D 27-point stencil on an N^3 grid (27 nnz/row: generic kernel)")
    ap.add_argument("cfgs", nargs="*", default=["index_codes=1", "index_codes=0", "dev.generic_spmv=1"])
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
    lib = pkg._lib.load()
    dtype = {"f32": np.float32, "f64": np.float64, "c64": np.complex64, "c128": np.complex128}[args.dtype]
    nx, ny, nz = (int(v) for v in args.grid.split("x"))
    n = nx * ny * nz
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    if args.helm:
        if np.dtype(dtype).kind != "c":
            raise SystemExit("--helm: the finite-element Helmholtz matrix is complex (--dtype c64 / c128)")
        n = args.helm ** 2
        indptr, indices, data = pkg.generators.helm_fe_var(ctx, args.helm, 12.0, None, 0.15, dtype=dtype)
    elif args.stencil27:
        import scipy.sparse as sp
        m = args.stencil27
        t1 = sp.diags([np.ones(m - 1), np.ones(m), np.ones(m - 1)], [-1, 0, 1], format="csr")
        A = sp.kron(sp.kron(t1, t1, format="csr"), t1, format="csr")
        A = (sp.identity(m ** 3, format="csr") * 27.0 - A * 0.5).tocsr()
        A.sort_indices()
        n = m ** 3
        indptr, indices = torch.from_numpy(A.indptr.astype(np.int32)).to(dev), torch.from_numpy(A.indices.astype(np.int32)).to(dev)
        data = torch.from_numpy(A.data.astype(dtype)).to(dev)
    else:
        indptr, indices, data = pkg.generators.laplace3d(ctx, nx, ny, nz, dtype=dtype)
    nnz = int(indices.numel())
    tdt = pkg.generators.torch_dtype(dtype)
    b = torch.full((n,), 5.0, dtype=tdt, device=dev)
    xs = torch.rand(n, dtype=torch.float64, device=dev).to(tdt)
    if np.dtype(dtype).kind == "c":
        b = b * (1 + 0.5j)
    ys = torch.empty(n, dtype=tdt, device=dev)
    torch.cuda.synchronize()
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    defaults = {"dev.generic_spmv": 0, "spmv_nt": -1, "dev.spmv_grid": 0, "vec_grid": 0, "spmv_cycle": 64, "dev.spmv_chunked": 1, "dev.spmv_chunk_kb": 0, "dev.spmv_slice_kb": 0, "dev.no_fold_alpha": 0, "dev.spmv_unroll": 0, "vec_nt": -1, "index_codes": 1, "index_codes16": 1, "dev.value_codes": 1, "dev.vc_pipe": 1, "dev.joint_codes": 1}
    solvers = []
    for cfg in args.cfgs:
        kv = dict(defaults)
        for pair in cfg.split(","):
            if pair:
                k, v = pair.split("=")
                kv[k] = int(v)
        for k, v in kv.items():
            pkg._lib.check(lib.cgamd_tune(k.encode(), v))
        s = pkg.Solver(ctx, n, nnz, data, indptr, indices, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
        solvers.append((cfg, kv, s))
    res = {cfg: {"spmv": [], "iter": []} for cfg, _, _ in solvers}
    ref = None
    for rnd in range(args.rounds):
        for cfg, kv, s in solvers:
            for k, v in kv.items():
                lib.cgamd_tune(k.encode(), v)       # (every solver runs with the snapshot taken when it was created)
            for _ in range(3):
                s.spmv(xs, ys, fused_dot=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(args.reps):
                s.spmv(xs, ys, fused_dot=True)
            e1.record(ext)
            e1.synchronize()
            res[cfg]["spmv"].append(e0.elapsed_time(e1) / args.reps * 1e3)
            if rnd == 0 and not (kv.get("spmv_dbg", 0) & 3):
                y = ys.clone()
                if ref is None:
                    ref = y
                else:
                    err = (y - ref).abs().max().item() / ref.abs().max().item()
                    assert err < (1e-12 if np.dtype(dtype).itemsize // (2 if np.dtype(dtype).kind == "c" else 1) == 8 else 1e-5), (cfg, err)
            if kv.get("spmv_dbg", 0) & 3:
                res[cfg]["iter"].append(1.0)
                continue
            # whole CG iterations (graph is captured with the tuning active at first iterate())
            s.set_rhs(b, None, on_device=True)
            s.iterate(10)
            ctx.synchronize()
            t0 = time.perf_counter()
            s.iterate(args.iters)
            ctx.synchronize()
            res[cfg]["iter"].append((time.perf_counter() - t0) / args.iters * 1e6)
    print(f"grid {args.grid} {args.dtype}: n={n} nnz={nnz}; percentages are MOVED bytes (each configuration's own index bytes and "
          f"vector passes) against 8 TB/s; csr = the reference's byte model as an effective rate")
    for cfg, _, sv in solvers:
        sp, it = np.array(res[cfg]["spmv"]), np.array(res[cfg]["iter"])
        sb, ib, cb = sv.spmv_moved_bytes, sv.iter_moved_bytes, sv.spmv_bytes
        print(f"{cfg:45s} spmv us med {np.median(sp):8.1f} min {sp.min():8.1f}  ({sb / np.median(sp) / 1e3:7.1f} GB/s moved = "
              f"{sb / np.median(sp) / 1e3 / 80:5.1f}% of 8TB/s; csr {cb / np.median(sp) / 1e3:7.1f} GB/s) | iter us med {np.median(it):8.1f} "
              f"min {it.min():8.1f} ({1e6 / np.median(it):7.1f} it/s, {ib / np.median(it) / 1e3 / 80:5.1f}% moved)")


if __name__ == "__main__":
    main()
