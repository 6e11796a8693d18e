#!/usr/bin/env python3
"""In-process A/B of tuning knobs on the resident loop (as_prec shape: helmFE_var(128) complex64, 16384 rows).
usage: resident_ab.py [--nrhs 9] [--iters 2560] [--rounds 5] knob=value[,knob=value] ..."""
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
ap.add_argument("--nrhs", type=int, default=9)
ap.add_argument("--iters", type=int, default=2560)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dtype", default="c64")
ap.add_argument("--N", type=int, default=128)
ap.add_argument("cfgs", nargs="+")
args = ap.parse_args()
pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
lib = pkg._lib.load()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
dt = {"c64": np.complex64, "c128": np.complex128, "f64": np.float64, "f32": np.float32}[args.dtype]
if args.N < 0:          # 3-D 7-point Laplacian (-N)^3
    ip, ix, da = pkg.generators.laplace3d(ctx, -args.N, -args.N, -args.N, dtype=dt)
elif args.dtype in ("c64", "c128"):
    ip, ix, da = pkg.generators.helm_fe_var(ctx, args.N, 12.0, None, 0.15, dtype=dt)
else:
    ip, ix, da = pkg.generators.poisson2d(ctx, args.N, dtype=dt)
n = int(ip.numel()) - 1
nnz = int(ix.numel())
tdt = pkg.generators.torch_dtype(dt)
b = torch.full((n * args.nrhs,), 5.0, dtype=tdt, device=dev)
solvers = []
for cfg in args.cfgs:
    kv = [x.split("=") for x in cfg.split(",") if "=" in x]
    for k, v in kv:
        pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))
    s = pkg.Solver(ctx, n, nnz, da, ip, ix, args.nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dt)
    for k, v in kv:
        pkg._lib.check(lib.cgamd_tune(k.encode(), {"resident": 1, "resident_min": 8, "dev.resident_window": 1, "resident_wide": 1}.get(k, 0)))
    solvers.append((cfg, s, []))
torch.cuda.synchronize()
for rnd in range(args.rounds):
    for cfg, s, ts in solvers:
        s.set_rhs(b, None, on_device=True)
        s.iterate(64)
        ctx.synchronize()
        t0 = time.perf_counter()
        s.iterate(args.iters)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / args.iters * 1e6)
for cfg, s, ts in solvers:
    print(json.dumps({"cfg": cfg, "n": n, "nrhs": args.nrhs, "dtype": args.dtype, "launches": lib.cgamd_solver_loop_launches(s.handle),
                      "us_per_iter_min": round(min(ts), 3), "us_per_iter_all": [round(t, 3) for t in ts]}), flush=True)
