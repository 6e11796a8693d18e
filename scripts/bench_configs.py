#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs 2-4 (parity-test cases, not the bench line):
prints one JSON line per config: CG it/s, SpMV/SpMM GB/s (algorithmic bytes)."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
lib = pkg._lib.load()
ctx = pkg.Context(0)
dev = torch.device("cuda", 0)
ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
NPD = {"f32": np.float32, "f64": np.float64, "c64": np.complex64, "c128": np.complex128}


def run(name, indptr, indices, data, dtype, nrhs, iters=200, reps=30, b1=None):
    n = indptr.numel() - 1
    nnz = indices.numel()
    tdt = pkg.generators.torch_dtype(dtype)
    torch.cuda.synchronize()        # (matrices built with torch ops live on torch's stream; the solver reads them on its own)
    s = pkg.Solver(ctx, n, nnz, data, indptr, indices, nrhs, flags=pkg._lib.MATRIX_ON_DEVICE | (pkg._lib.NO_GRAPH if "nograph" in sys.argv else 0), dtype=dtype)
    # b1: one right-hand side from the device generators (config 3: rhsA(500, 12)), repeated; else the CLI's b = 5 (main.c:44)
    b = b1.repeat(nrhs) if b1 is not None else torch.full((n * nrhs,), 5.0, dtype=tdt, device=dev)
    torch.cuda.synchronize()
    s.set_rhs(b, None, on_device=True)
    s.iterate(20)
    ctx.synchronize()
    t0 = time.perf_counter()
    s.iterate(iters)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    xs = torch.ones(n * nrhs, dtype=tdt, device=dev)
    ys = torch.empty(n * nrhs, dtype=tdt, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        s.spmv(xs, ys, fused_dot=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(reps):
        s.spmv(xs, ys, fused_dot=True)
    e1.record(ext)
    e1.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    # fractions are priced on the bytes the handle's kernels move (its own index bytes / vector passes); the reference's CSR byte
    # model is printed as an effective rate only
    sb, ib, cb = s.spmv_moved_bytes, s.iter_moved_bytes, s.spmv_bytes
    launches = lib.cgamd_solver_loop_launches(s.handle)
    out = {"config": name, "n": n, "nnz": nnz, "dtype": np.dtype(dtype).name, "nrhs": nrhs, "index_codes": s.index_codes,
           "launches_per_iteration": launches, "cg_it_per_s": iters / dt, "us_per_iter": dt / iters * 1e6,
           "spmv_us": us, "spmv_moved_gbs": sb / us / 1e3, "spmv_pct_of_8tbs": 100 * sb / us / 1e3 / 8000,
           "spmv_effective_csr_gbs": cb / us / 1e3}
    if launches >= 2:       # the resident loops stream nothing per iteration: a streaming fraction would mean nothing there
        out["cg_iter_pct_of_8tbs"] = 100 * ib * iters / dt / 8e12
    print(json.dumps(out), flush=True)
    s.close()


which = [w for w in sys.argv[1:] if "=" not in w and w != "nograph"] or ["c2", "c3", "c4", "m32", "c2f32", "c3c64"]
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        pkg._lib.check(lib.cgamd_tune(k.encode(), int(v)))
if "c2" in which:
    ip, ix, da = pkg.generators.poisson2d(ctx, 1000, dtype=np.float64)
    run("C2 2D 5-pt N=1M f64", ip, ix, da, np.float64, 1)
if "c2f32" in which:
    ip, ix, da = pkg.generators.poisson2d(ctx, 1000, dtype=np.float32)
    run("C2 2D 5-pt N=1M f32", ip, ix, da, np.float32, 1)
if "c3" in which or "c3c64" in which:
    N = 500          # helmFE_var(N=500, omega=12, C=1, rho=0.15), generated on the device (csrc/generators.hip)
    if "c3" in which:
        ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex128)
        run("C3 Helmholtz FE N=250k c128, b = rhsA(500, 12)", ip, ix, da, np.complex128, 1, b1=pkg.generators.rhsA(ctx, N, 12.0, dtype=np.complex128))
    if "c3c64" in which:
        ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
        bA = pkg.generators.rhsA(ctx, N, 12.0, dtype=np.complex64)
        run("C3 Helmholtz FE N=250k c64 (reference dtype), b = rhsA(500, 12)", ip, ix, da, np.complex64, 1, b1=bA)
        run("C3 Helmholtz FE N=250k c64 nrhs=9 (as_prec shape)", ip, ix, da, np.complex64, 9, iters=100, b1=bA)
if "c4" in which:
    ip, ix, da = pkg.generators.poisson2d(ctx, 1000, dtype=np.float64)
    run("C4 SpMM nrhs=32 N=1M f64", ip, ix, da, np.float64, 32, iters=50, reps=10)
if "c4" in which or "c4mfma" in which:
    # the same product on the matrix cores, row-major RHS block (transposes excluded: layout is kept by a caller)
    for dt, tdt_, nm in ((np.float64, torch.float64, "f64"), (np.float32, torch.float32, "f32")):
        ip, ix, da = pkg.generators.poisson2d(ctx, 1000, dtype=dt)
        n = 1000 * 1000
        s = pkg.Solver(ctx, n, ix.numel(), da, ip, ix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=dt)
        for nrhs in (32, 16):
            xs = torch.rand(n * nrhs, dtype=tdt_, device=dev)
            ys = torch.empty(n * nrhs, dtype=tdt_, device=dev)
            torch.cuda.synchronize()
            for _ in range(3):
                s.spmm_rowmajor(xs, ys, nrhs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ext)
            for _ in range(10):
                s.spmm_rowmajor(xs, ys, nrhs)
            e1.record(ext)
            e1.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            V = np.dtype(dt).itemsize
            sb = ix.numel() * (V + 4) + (n + 1) * 4 + 2 * n * V * nrhs
            print(json.dumps({"config": f"C4 SpMM MFMA row-major nrhs={nrhs} N=1M {nm}", "spmm_us": us, "spmm_gbs": sb / us / 1e3,
                              "spmm_pct_of_8tbs": 100 * sb / us / 1e3 / 8000}), flush=True)
        s.close()
if "irregular" in which:
    # Patterns that are NOT a stencil: generated on the device, no download.  Each line: SpMV time, the rate of
    # the bytes the kernel moves (its own index bytes), the effective CSR rate, and which index form the handle chose
    # (index_codes: 0 = aCols, 1..256 = one-byte codes, 65536 = 16-bit block-relative columns).
    def csr_from_coo(rows, cols, vals, n):
        key = rows.to(torch.int64) * n + cols.to(torch.int64)
        key, order = torch.sort(key)
        keep = torch.ones_like(key, dtype=torch.bool)
        keep[1:] = key[1:] != key[:-1]                      # (generated patterns may repeat an entry: first one wins)
        key, order = key[keep], order[keep]
        r = (key // n).to(torch.int64)
        ip = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ip[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
        return ip.to(torch.int32), (key % n).to(torch.int32), vals[order].contiguous()

    def spd_from_pattern(rows, cols, n, gen):
        # symmetric pattern, off-diagonal values in (-1, -0.5), diagonal = 1e-3 - sum of the row's off-diagonals: SPD
        rr, cc = torch.cat([rows, cols]), torch.cat([cols, rows])
        off = rr != cc
        rr, cc = rr[off], cc[off]
        lo, hi = torch.minimum(rr, cc), torch.maximum(rr, cc)
        h = (lo.to(torch.int64) * 2654435761 + hi.to(torch.int64) * 40503) % 1000003          # the same value for (i, j) and (j, i)
        v = -(0.5 + 0.5 * (h.to(torch.float64) / 1000003.0))
        ip, ix, da = csr_from_coo(rr, cc, v, n)
        rowsum = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(n, device=dev), (ip[1:] - ip[:-1]).long()), da)
        d = torch.arange(n, device=dev)
        return csr_from_coo(torch.cat([torch.repeat_interleave(torch.arange(n, device=dev), (ip[1:] - ip[:-1]).long()), d]),
                            torch.cat([ix.long(), d]), torch.cat([da, 1e-3 - rowsum]), n)

    gen = torch.Generator(device=dev).manual_seed(1234)
    # (1) the headline matrix under a random symmetric permutation: every row's neighbours are anywhere
    ip, ix, da = pkg.generators.laplace3d(ctx, 250, 200, 200, dtype=np.float64)
    n = ip.numel() - 1
    perm = torch.randperm(n, device=dev, generator=gen)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (ip[1:] - ip[:-1]).long())
    ip2, ix2, da2 = csr_from_coo(perm[rows], perm[ix.long()], da, n)
    del ip, ix, da, rows
    run("irregular: 10M 7-pt Laplacian, random symmetric permutation f64", ip2, ix2, da2, np.float64, 1, iters=50, reps=10)
    del ip2, ix2, da2
    torch.cuda.empty_cache()
    # (2) the report's m_t1 shape (n = 97 578, ~100 entries per row): random columns inside a band of +-5000
    n, per = 100_000, 50
    rows = torch.repeat_interleave(torch.arange(n, device=dev), per)
    cols = rows + torch.randint(-5000, 5001, (n * per,), device=dev, generator=gen)
    cols = torch.where(cols < 0, -cols, torch.where(cols > n - 1, 2 * (n - 1) - cols, cols))      # reflected at the ends (clamping would make
                                                                                                  # rows 0 and n - 1 hubs of thousands of entries)
    ip, ix, da = spd_from_pattern(rows, cols, n, gen)
    run("irregular: banded random n=100k, ~100 per row (m_t1 shape) f64", ip, ix, da, np.float64, 1, iters=200, reps=30)
    # (3) the report's parabolic_fem shape (n = 525 825, ~7 per row): a P1 mesh pattern whose nodes are renumbered at random inside
    # windows of 4096 -- the offsets of an unstructured mesh in a bandwidth-reducing order: hundreds of distinct ones, bounded band
    N = 725
    ipf, ixf, daf = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
    n = N * N
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (ipf[1:] - ipf[:-1]).long())
    win = 4096
    keys = (torch.arange(n, device=dev) // win).to(torch.float64) + torch.rand(n, device=dev, dtype=torch.float64, generator=gen) * 0.999
    order = torch.argsort(keys)
    perm = torch.empty_like(order)
    perm[order] = torch.arange(n, device=dev)
    ip, ix, da = spd_from_pattern(perm[rows], perm[ixf.long()], n, gen)
    del ipf, ixf, daf
    run("irregular: mesh-like n=525k, ~7 per row, locally renumbered (parabolic_fem shape) f64", ip, ix, da, np.float64, 1, iters=200, reps=30)
if "asprec" in which:
    # the reference's own sub-domain solve (as_prec, p_h-PY_C-CL.py:1918-1953): complex64, ~16k rows (helmFE_var(128) has the
    # pattern and size of local_rect for W_s + 2 ol = 128), n_my = 9 right-hand sides, CGMaxIT = 256 fixed iterations
    N = 128
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
    for nrhs in (9, 1):
        run(f"as_prec shape: helmFE_var(128) c64 n=16384 nrhs={nrhs}", ip, ix, da, np.complex64, nrhs, iters=2560, reps=30)
if "mid" in which:
    # between the reference's sub-domain size and config 3: chip-wide resident groups, one per right-hand side
    ip, ix, da = pkg.generators.poisson2d(ctx, 300, dtype=np.float64)
    run("mid-size: 2D 5-pt 90000 rows f64 nrhs=9", ip, ix, da, np.float64, 9, iters=1000, reps=20)
    ip, ix, da = pkg.generators.helm_fe_var(ctx, 200, 12.0, None, 0.15, dtype=np.complex64)
    run("mid-size: helmFE_var(200) c64 40000 rows nrhs=9", ip, ix, da, np.complex64, 9, iters=1000, reps=20)
if "report" in which:
    # context only: the one matrix of the upstream report that can be regenerated offline --
    # helm_fem: complex, n = 16 384, nnz = 113 666 = helmFE_var(N=128); the report ran 5000 iterations in fp32 complex and
    # counts 8 nnz + 40 n flops per iteration (its Table II): 2.390 GFLOPS on an RTX 2080S, 0.351 on an i5-8250U
    N = 128
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
    n, nnz = N * N, int(ix.numel())
    s = pkg.Solver(ctx, n, nnz, da, ip, ix, 1, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.complex64)
    b = torch.full((n,), 5.0, dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()
    s.set_rhs(b, None, on_device=True)
    s.iterate(200)
    ctx.synchronize()
    t0 = time.perf_counter()
    s.iterate(5000)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    flops = 8 * nnz + 40 * n
    print(json.dumps({"config": "report helm_fem equivalent: helmFE_var(128) c64, n=16384, nnz=%d, 5000 iterations" % nnz,
                      "cg_it_per_s": 5000 / dt, "us_per_iter": dt / 5000 * 1e6, "gflops_report_model": flops * 5000 / dt / 1e9,
                      "published_2080S_gflops": 2.390}), flush=True)
    s.close()
if "c5" in which:
    # BASELINE config 5's system whole on one GPU (N = 99.9M, 8.4 GB of CSR), and one rank's 1/8 z-slab of it
    ip, ix, da = pkg.generators.laplace3d(ctx, 464, 464, 464, dtype=np.float64)
    run("C5 3D 7-pt 464^3 N=99.9M f64, one GPU", ip, ix, da, np.float64, 1, iters=40, reps=8)
    del ip, ix, da
    torch.cuda.empty_cache()
if "c5" in which or "c5slab" in which:
    ip, ix, da = pkg.generators.laplace3d(ctx, 464, 464, 58, dtype=np.float64)
    run("C5-slab 464x464x58 (one rank of 8) N=12.49M f64, plain solver", ip, ix, da, np.float64, 1, iters=100, reps=20)
    del ip, ix, da
    torch.cuda.empty_cache()
if "m32" in which:
    ip, ix, da = pkg.generators.laplace3d(ctx, 250, 200, 200, dtype=np.float32)
    run("M 3D 7-pt N=10M f32", ip, ix, da, np.float32, 1)
