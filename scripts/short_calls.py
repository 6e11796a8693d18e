#!/usr/bin/env python3
"""Wall time of short iterate(K) calls: chip-wide resident loop against the launched loops (break-even call length)."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
lib = pkg._lib.load()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
for name, N, dt in (("C2 f64 1M", 1000, np.float64), ("C3 c64 250k", 500, np.complex64), ("f64 90k", 300, np.float64)):
    if dt == np.complex64:
        ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=dt)
    else:
        ip, ix, da = pkg.generators.poisson2d(ctx, N, dtype=dt)
    n = N * N
    b = torch.full((n,), 5.0, dtype=pkg.generators.torch_dtype(dt), device=dev)
    for wide in (1, 0):
        pkg._lib.check(lib.cgamd_tune(b"resident_wide", wide))
        pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 1))
        s = pkg.Solver(ctx, n, int(ix.numel()), da, ip, ix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dt)
        pkg._lib.check(lib.cgamd_tune(b"resident_wide", 1))
        pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))
        s.set_rhs(b, None, on_device=True)
        s.iterate(64); ctx.synchronize()
        row = []
        for K in (4, 8, 16, 32, 64, 256):
            best = 1e9
            for rep in range(5):
                t0 = time.perf_counter(); s.iterate(K); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
            row.append((K, round(best * 1e6, 1)))
        print(name, "wide" if wide else "launched", lib.cgamd_solver_loop_launches(s.handle), row, flush=True)
        s.close()
