#!/usr/bin/env python3
"""In-process A/B of the row-major matrix-core SpMM (csrc/rowmajor.hip) and of the multi-RHS CG loop around it.
usage: python scripts/spmm_ab.py [--N 1000] [--dtype f64] [--nrhs 32] [--helm 0] cfg1 cfg2 ...
  cfg = comma-separated key=value tuning pairs (cgamd_tune), e.g.  dev.spmm_wgs=64  dev.spmm_wgs=32,dev.spmm_ynt=1 ;  "cg" in a cfg
  also times the CG loop (it/s)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1000)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--nrhs", type=int, default=32)
    ap.add_argument("--helm", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("cfgs", nargs="*", default=["dev.spmm_wgs=64"])
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
    lib = pkg._lib.load()
    dtype = {"f32": np.float32, "f64": np.float64, "c64": np.complex64}[args.dtype]
    ctx = pkg.Context(0)
    dev = torch.device("cuda", 0)
    if args.helm:
        if np.dtype(dtype).kind != "c":
            raise SystemExit("--helm: the finite-element Helmholtz matrix is complex (--dtype c64)")
        n = args.helm ** 2
        ip, ix, da = pkg.generators.helm_fe_var(ctx, args.helm, 12.0, None, 0.15, dtype=dtype)
    else:
        n = args.N * args.N
        ip, ix, da = pkg.generators.poisson2d(ctx, args.N, dtype=dtype)
    nnz = int(ix.numel())
    tdt = pkg.generators.torch_dtype(dtype)
    V = np.dtype(dtype).itemsize
    nrhs = args.nrhs
    sb = nnz * (V + 4) + (n + 1) * 4 + 2 * n * V * nrhs
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    xs = torch.rand(n * nrhs, dtype=torch.float64, device=dev).to(tdt)
    ys = torch.empty(n * nrhs, dtype=tdt, device=dev)
    b = torch.full((n * nrhs,), 5.0, dtype=tdt, device=dev)
    torch.cuda.synchronize()
    defaults = {}
    res = {c: {"spmm_us": [], "cg_it_s": []} for c in args.cfgs}
    for rnd in range(args.rounds):
        for cfg in args.cfgs:
            pairs = [kv.split("=") for kv in cfg.split(",") if "=" in kv]
            for k, v in pairs:
                pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))
            s = pkg.Solver(ctx, n, nnz, da, ip, ix, nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dtype)
            for _ in range(3):
                s.spmm_rowmajor(xs, ys, nrhs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(args.reps):
                s.spmm_rowmajor(xs, ys, nrhs)
            e1.record(ext)
            e1.synchronize()
            res[cfg]["spmm_us"].append(e0.elapsed_time(e1) / args.reps * 1e3)
            if "cg" in cfg.split(","):
                s.set_rhs(b, None, on_device=True)
                s.iterate(10)
                ctx.synchronize()
                t0 = time.perf_counter()
                s.iterate(40)
                ctx.synchronize()
                res[cfg]["cg_it_s"].append(40 / (time.perf_counter() - t0))
            s.close()
            for k, v in pairs:      # back to defaults (0 = auto for every knob used here except spmm_rowmajor)
                pkg._lib.check(lib.cgamd_tune(k.encode(), {"spmm_rowmajor": 1, "dev.spmm_ynt": -1}.get(k, 0)))
    for cfg in args.cfgs:
        us = min(res[cfg]["spmm_us"])
        out = {"cfg": cfg, "n": n, "nnz": nnz, "dtype": args.dtype, "nrhs": nrhs, "spmm_us_min": round(us, 1),
               "spmm_us_all": [round(u, 1) for u in res[cfg]["spmm_us"]], "pct_of_8tbs": round(100 * sb / us / 1e3 / 8000, 1)}
        if res[cfg]["cg_it_s"]:
            out["cg_it_s_max"] = round(max(res[cfg]["cg_it_s"]), 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
