#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: a known-bytes streaming read (vdot over two 800 MB
vectors = calibration of FETCH_SIZE on this access pattern), a known-bytes read+write (axpy), then the
solver's fused SpMV and a few CG iterations on the N=10M system.  Also prints HIP-event timings."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
lib = pkg._lib.load()
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
kern = pkg.load_and_build_kernels(ctx, 1)
queue = pkg.CommandQueue(ctx)


def timed(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        fn()
    e1.record(ext)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


NB = 100_000_000 if os.environ.get("PMC_CALIB", "1") == "1" else 1000
a = torch.rand(NB, dtype=torch.float64, device=dev)
b = torch.rand(NB, dtype=torch.float64, device=dev)
res = torch.zeros(4, dtype=torch.float64, device=dev)
al = torch.full((1,), 0.5, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
us = timed(lambda: kern["vdot"](queue, a, b, res, NB), 5)
print(f"CALIB vdot read {2 * NB * 8} B: {us:.1f} us = {2 * NB * 8 / us / 1e3:.1f} GB/s")
us = timed(lambda: kern["axpy"](queue, a, b, al, 1, NB), 5)
print(f"CALIB axpy read {2 * NB * 8} B write {NB * 8} B: {us:.1f} us = {3 * NB * 8 / us / 1e3:.1f} GB/s")
del a, b

nx, ny, nz = 250, 200, 200
n = nx * ny * nz
indptr, indices, data = pkg.generators.laplace3d(ctx, nx, ny, nz, dtype=np.float64)
nnz = int(indices.numel())
s = pkg.Solver(ctx, n, nnz, data, indptr, indices, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.float64)
xs = torch.rand(n, dtype=torch.float64, device=dev)
ys = torch.empty(n, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
us = timed(lambda: s.spmv(xs, ys, fused_dot=True), 10)
print(f"SPMV fused algorithmic {s.spmv_bytes} B: {us:.1f} us = {s.spmv_bytes / us / 1e3:.1f} GB/s")
bvec = torch.full((n,), 5.0, dtype=torch.float64, device=dev)
s.set_rhs(bvec, None, on_device=True)
s.iterate(10)
ctx.synchronize()
print("done")
