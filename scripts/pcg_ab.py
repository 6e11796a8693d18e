#!/usr/bin/env python3
"""Jacobi-PCG iteration time: chip-wide resident loop against the launched four-launch loop of the same handle.
usage: pcg_ab.py [--N 128] [--dtype c64] [--nrhs 1] [--iters 2000]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--dtype", default="c64")
ap.add_argument("--nrhs", type=int, default=1)
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
lib = pkg._lib.load()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
dt = {"c64": np.complex64, "c128": np.complex128, "f64": np.float64, "f32": np.float32}[args.dtype]
if args.dtype in ("c64", "c128"):
    ip, ix, da = pkg.generators.helm_fe_var(ctx, args.N, 12.0, None, 0.15, dtype=dt)
else:
    ip, ix, da = pkg.generators.poisson2d(ctx, args.N, dtype=dt)
n, nnz = int(ip.numel()) - 1, int(ix.numel())
rows = torch.repeat_interleave(torch.arange(n, device=dev), (ip[1:] - ip[:-1]).long())
diag = torch.zeros(n, dtype=da.dtype, device=dev)
diag[rows[ix.long() == rows]] = da[ix.long() == rows]
m = (1.0 / diag).contiguous()
b = torch.full((n * args.nrhs,), 5.0, dtype=da.dtype, device=dev)
torch.cuda.synchronize()
for name, wide_min in (("resident", 16), ("launched", 1 << 20)):
    pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", wide_min))
    s = pkg.Solver(ctx, n, nnz, da, ip, ix, args.nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dt)
    pkg._lib.check(lib.cgamd_solver_set_preconditioner(s.handle, pkg._lib.ptr(m), 1))
    ts = []
    for _ in range(args.rounds):
        s.set_rhs(b, None, on_device=True)
        s.iterate(64)
        ctx.synchronize()
        t0 = time.perf_counter()
        s.iterate(args.iters)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / args.iters * 1e6)
    print(json.dumps({"loop": name, "n": n, "nrhs": args.nrhs, "dtype": args.dtype, "launches": lib.cgamd_solver_loop_launches(s.handle),
                      "us_per_iter_min": round(min(ts), 3), "finite": bool(np.all(np.isfinite(s.history()[:200])))}), flush=True)
    s.close()
pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))
