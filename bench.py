#!/usr/bin/env python3
"""Headline benchmark: CG iterations/s + SpMV effective HBM GB/s on the synthetic N=10M 7-point system.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" is one CG iteration (the reference's loop body, clcg.c:297-419) on the 3-D 7-point Laplacian
250x200x200 (N=10 000 000, nnz=69 720 000, fp64), matrix and vectors resident in HBM before the timed
region.  With N>1 ranks the SAME system is row-partitioned into N contiguous z-slabs (strong scaling, as the
north star asks: "iterations/sec at 8 GPUs vs 1 GPU on the 10M-row system"): neighbour halo exchange +
2 scalar all-reduces per iteration, by peer-to-peer mailbox writes over xGMI inside the iteration's four kernel launches
(RCCL send/recv + all-reduce as the fallback); every candidate loop is validated against the single-GPU residual history.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "conjugate-gradient-pyopencl_amd"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", type=str, default="250x200x200", help="global grid nx x ny x nz")
    ap.add_argument("--dist-skip", type=str, default="", help="N>1: comma list of loop candidates to skip (slab,p2p2-sr,p2p4,p2p4+graph,p2p,p2p+graph,rccl-sr,rccl-sr+graph,rccl+graph,rccl)")
    ap.add_argument("--dtype", type=str, default="f64", choices=["f32", "f64", "c64", "c128"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-system", action="store_true", help="skip the small-system extra (reference call shape, resident loop)")
    ap.add_argument("--unfused", action="store_true", help="reference op structure (6 kernels/iteration)")
    ap.add_argument("--no-graph", action="store_true", help="plain stream launches instead of hipGraph replay (A/B)")
    ap.add_argument("--spmv-reps", type=int, default=20)
    return ap.parse_args()


NP_DTYPE = {"f32": np.float32, "f64": np.float64, "c64": np.complex64, "c128": np.complex128}


def host_cores():
    """cores this process may really use: cgroup quota if any, affinity mask, and at most 16 -- a one-GPU box
    of the pool owns a 16-core CPU share even though it sees every core of the host."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("CG_BASELINE_THREADS", "16"))))


def pmc_traffic(kernel="spmv_rowblock_kernel"):
    """(HBM bytes per launch of the dominant kernel, where that number comes from).  PMC counters need rocprofv3 around
    the process, so the figure is read from the committed counter passes of the same command (scripts/pmc_traffic.py);
    the newest round's file FOR THE KERNEL FORM THAT RUNS wins.  (None, reason) when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    for f in reversed(files):
        try:
            j = json.load(open(f))
            if j.get("kernel") != kernel:
                continue
            return j["hbm_bytes_per_launch"], (f"profiles/{os.path.basename(f)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                               f"over bench.py, {j.get('launches_averaged', '?')} launches averaged; not measured in this run)")
        except Exception:
            continue
    return None, "no committed PMC pass of " + kernel + " found"


def cpu_baseline(nx, ny, nz, dtype, budget_s=10.0, batch=50):
    """The CPU oracle (C restatement of the reference op structure clcg.c:298-416: spmv, vdot, axpy, axpy, vdot, aypx,
    host-resident scalars, no fusion; OpenMP) on the FULL system of the GPU run -- same matrix, same b, same dtype --
    for about `budget_s` seconds of host work: solves of `batch` iterations from x0 = 0, repeated."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cg_numpy
    import cg_oracle
    threads = cg_oracle.set_threads(host_cores())
    if np.dtype(dtype).kind == "f":
        indptr, indices, data = cg_oracle.laplace3d(nx, ny, nz, dtype=dtype)
    else:
        indptr, indices, data = cg_numpy.laplace3d(nx, ny, nz, dtype=dtype)
    n = nx * ny * nz
    b = np.full(n, 5.0, dtype=dtype)                   # main.c:44 convention
    cg_oracle.cg(indptr, indices, data, b, n_iterations=1, mode=cg_oracle.MODE_FAST)      # first touch / warm-up
    iters, hist = 0, None
    t0 = time.perf_counter()
    while iters == 0 or (time.perf_counter() - t0 < budget_s and iters < 100 * batch):
        _, hist = cg_oracle.cg(indptr, indices, data, b, n_iterations=batch, mode=cg_oracle.MODE_FAST)
        iters += batch
    dt = time.perf_counter() - t0
    # every solve also runs the setup (one SpMV, sub, copy, dot: clcg.c:255-292), ~0.6 of an iteration per `batch`: not counted
    return {"value": iters / dt, "unit": "CG iterations/s", "cores": threads, "kind": "port",
            "sample": f"oracle/cg_oracle.c (OpenMP, {threads} threads, reference op structure) {iters} iterations "
                      f"({iters // batch} solves of {batch} from x0=0, setup included in the time) on the full {nx}x{ny}x{nz} "
                      f"system ({n} rows, {len(indices)} non-zeros, {np.dtype(dtype).name}) in {dt:.2f} s; "
                      f"delta_{batch}/delta_0 = {abs(hist[-1, 0]) / abs(hist[0, 0]):.3e}"}


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N rank processes ourselves.

    Runs BEFORE anything in this process has touched HIP (torch is not even imported here): the ranks are fresh children
    of `python -m torch.distributed.run`, this process only relays rank 0's JSON line and the exit code -- a process
    that has initialised the GPU is never re-executed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line_json = None
    for line in proc.stdout:
        if line.lstrip().startswith("{") and '"metric"' in line:
            line_json = line.strip()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if line_json is not None:
        print(line_json, flush=True)
    if rc == 0 and line_json is None:
        sys.stderr.write("bench.py: the rank processes exited 0 without a result line\n")
        rc = 1
    return rc


def main():
    args = parse()
    # multi-process GPU work on this pool needs dmabuf IPC; set before any HIP call of this process
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("CG_FORCE_DIST", "0") != "1":
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        args.gpus = world
    ndev = torch.cuda.device_count()
    shared = world > max(ndev, 1)     # rehearsal of N ranks on fewer GPUs (a 1-GPU box); never the case on the 8-GPU node
    if local_rank >= ndev:
        local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("CG_FORCE_DIST", "0") == "1":
        # RCCL refuses two ranks on one device: ranks that share a GPU bootstrap over gloo and run the peer-to-peer loops only
        backend = os.environ.get("CG_DIST_BACKEND", "gloo" if shared else "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    pkg = importlib.import_module(PKG)
    lib = pkg._lib
    tune = os.environ.get("CG_TUNE", "")
    if shared and "dev.resident_lock" not in tune:
        # ranks of this job share a GPU: their slab launches must run at the same time (no per-GPU serialisation of resident launches)
        tune = ",".join(filter(None, [tune, "dev.resident_lock=0"]))
    if shared and "vec_grid" not in tune:
        # ranks sharing one GPU must leave each other's spinning work-groups room to run
        per_dev = (world + max(ndev, 1) - 1) // max(ndev, 1)
        tune = ",".join(filter(None, [tune, f"vec_grid={max(64, 1536 // per_dev)}"]))
    for pair in filter(None, tune.split(",")):      # experiment knobs, e.g. CG_TUNE=vec_grid=512
        k, v = pair.split("=")
        lib.check(lib.load().cgamd_tune(k.encode(), int(v)))
    dtype = NP_DTYPE[args.dtype]
    nx, ny, nz = (int(v) for v in args.grid.split("x"))
    ctx = pkg.Context(local_rank)

    if world == 1 and os.environ.get("CG_FORCE_DIST", "0") != "1":
        result = bench_single(args, pkg, ctx, torch, dev, nx, ny, nz, dtype)
    else:
        from importlib import import_module
        distmod = import_module(PKG + ".dist")
        result = distmod.bench_strong_scaling(args, pkg, ctx, torch, dist, dev, nx, ny, nz, dtype, rank, world)
        if shared:
            result["rehearsal"] = f"{world} ranks share {max(ndev, 1)} GPU(s): bootstrap over gloo, peer-to-peer loops only, not a scaling number"

    if rank == 0:
        if world == 1 and not args.no_small_system:
            try:
                result["small_system"] = small_system_extra(pkg, ctx, torch, dev)
            except Exception as e:   # a reported extra, never the measured path
                result["small_system"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                result["cpu_baseline"] = cpu_baseline(nx, ny, nz, dtype)
            except Exception as e:   # the baseline is a reported extra, never the measured path
                result["cpu_baseline"] = {"value": None, "unit": "CG iterations/s", "cores": os.cpu_count(),
                                          "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def small_system_extra(pkg, ctx, torch, dev):
    """Reported beside the headline, never part of `value`: the reference's own call shape (as_prec, p_h-PY_C-CL.py:1925-1950:
    one sub-domain matrix, 9 complex64 right-hand sides, fixed iterations) on the helmFE_var(128) matrix (16 384 rows), generated
    on the device; all iterations of a call run in one launch (csrc/resident.hip)."""
    import time
    N, nrhs, iters = 128, 9, 2560
    ip, ix, da = pkg.generators.helm_fe_var(ctx, N, 12.0, None, 0.15, dtype=np.complex64)
    s = pkg.Solver(ctx, N * N, int(ix.numel()), da, ip, ix, nrhs, flags=pkg._lib.MATRIX_ON_DEVICE, dtype=np.complex64)
    b = pkg.generators.rhsA(ctx, N, 12.0, dtype=np.complex64).repeat(nrhs)      # the drivers' right-hand side (helmFE_var.py:379-389)
    torch.cuda.synchronize()
    s.set_rhs(b, None, on_device=True)
    s.iterate(64)
    ctx.synchronize()
    t0 = time.perf_counter()
    s.iterate(iters)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    launches = pkg._lib.load().cgamd_solver_loop_launches(s.handle)
    h = s.history()
    s.close()
    return {"workload": "helmFE_var(128) complex64, 16384 rows x 9 right-hand sides (the reference's as_prec call shape)",
            "us_per_iteration": dt / iters * 1e6, "cg_it_per_s": iters / dt, "launches_per_iteration": launches,
            "finite_history": bool(np.all(np.isfinite(h[: iters + 65])))}


def rate_fields(spmv_moved, spmv_csr, spmv_ms, spmv_alone_ms, iter_moved, iter_csr, it_s, traffic, traffic_source, n_offsets, n_values=0, n_pairs=0):
    """The rate / roofline part of the single-GPU line (pure arithmetic: tests/test_bench_launch.py checks it without a GPU).

    `moved`: what the kernels that ran really move -- index bytes per non-zero as the SpMV reads them (1 with the one-byte
    column codes) and the loop's own vector passes (10).  Every FRACTION of the 8 TB/s peak is priced on
    these bytes, so none can exceed what the memory system delivers.  `csr`: the reference's CSR byte model of SURVEY 8(d)
    (12 B per fp64 non-zero, 14 vector passes of its unfused op structure) -- reported as an "effective" rate only."""
    spmv_gbs = spmv_moved / (spmv_ms * 1e-3) / 1e9
    alone_gbs = spmv_moved / (spmv_alone_ms * 1e-3) / 1e9
    return {
        "spmv_gbs": spmv_gbs, "spmv_pct_of_8tbs": 100.0 * spmv_gbs / HBM_PEAK_GBS,
        "spmv_back_to_back_gbs": alone_gbs, "spmv_back_to_back_pct_of_8tbs": 100.0 * alone_gbs / HBM_PEAK_GBS,
        "cg_iter_moved_bytes": iter_moved, "cg_iter_gbs": iter_moved * it_s / 1e9,
        "cg_iter_pct_of_8tbs": 100.0 * iter_moved * it_s / 1e9 / HBM_PEAK_GBS,
        "effective_csr": {"note": "reference CSR byte model (SURVEY 8d): 4 index bytes per non-zero, 14 vector passes; an effective "
                                  "rate for comparison with a solver that reads aCols, not a fraction of the peak",
                          "spmv_bytes": spmv_csr, "spmv_gbs": spmv_csr / (spmv_ms * 1e-3) / 1e9,
                          "cg_iter_bytes": iter_csr, "cg_iter_gbs": iter_csr * it_s / 1e9},
        "roofline": {"bound": "hbm",
                     "kernel": ("spmv_rowblock_vcp_kernel" if n_values > 0 else "spmv_rowblock_kernel") + " (CSR SpMV fused with d.q partials"
                               + (", column indices read as one-byte codes" if n_offsets > 0 else "")
                               + (", values as one-byte codes into the matrix's %d distinct entries" % n_values if n_values > 0 and not n_pairs else "")
                               + (", ONE code byte per non-zero naming one of the matrix's %d (offset, value) pairs" % n_pairs if n_pairs else "")
                               + (": no longer bound by bytes, see DESIGN.md" if n_values > 0 else "")
                               + "), in-loop average over the instrumented pass (HIP events on each dispatch)",
                     "achieved": spmv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_ratio": (traffic / spmv_moved) if traffic else None,
                     "moved_bytes_per_launch": spmv_moved, "avg_launch_ms": spmv_ms,
                     "index_codes": {"distinct_offsets": n_offsets, "index_bytes_per_nonzero": 2 if n_offsets == 65536 else 1 if n_offsets > 0 else 4},
                     "value_codes": {"distinct_values": n_values, "value_bytes_per_nonzero": 1 if n_values > 0 else None,
                                     "joint_pairs": n_pairs, "code_bytes_per_nonzero": 1 if n_pairs else 2 if n_values > 0 else None},
                     "effective_csr_bytes_per_launch": spmv_csr, "effective_csr_gbs": spmv_csr / (spmv_ms * 1e-3) / 1e9},
    }


def bench_single(args, pkg, ctx, torch, dev, nx, ny, nz, dtype):
    lib = pkg._lib
    n = nx * ny * nz
    indptr, indices, data = pkg.generators.laplace3d(ctx, nx, ny, nz, dtype=dtype)
    nnz = int(indices.numel())
    flags = lib.MATRIX_ON_DEVICE | (lib.UNFUSED if args.unfused else 0) | (lib.NO_GRAPH if args.no_graph else 0)
    solver = pkg.Solver(ctx, n, nnz, data, indptr, indices, 1, flags=flags, dtype=dtype)
    tdt = pkg.generators.torch_dtype(dtype)
    b = torch.full((n,), 5.0, dtype=tdt, device=dev)       # main.c:44: b = (r+1)*5, x0 = 0
    torch.cuda.synchronize()
    solver.set_rhs(b, None, on_device=True)
    solver.iterate(args.warmup)
    ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.iterate(args.steps)
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist = solver.history()
    it_s = args.steps / dt

    # ---- dominant kernel: the fused SpMV (+ d.q partials).  Its launch duration is measured live, IN the CG loop:
    # an instrumented pass of the same K iterations in which every SpMV dispatch carries its own HIP start/stop event
    # pair (hipExtLaunchKernelGGL on the solver's stream: the kernel's execution time, what rocprofv3's kernel trace
    # reports; plain launches -- the timed region above replays the same kernels from a hipGraph).
    spmv_ms = None
    if not args.unfused:
        spmv_ms, inst_iter_ms = solver.iterate_timed(args.steps)
        hist = solver.history()
    spmv_bytes = solver.spmv_bytes
    iter_bytes = solver.iter_bytes(fused=False)
    # for reference: the same kernel launched back to back on fixed vectors (x, y stay warm in the Infinity Cache)
    xs = torch.ones(n, dtype=tdt, device=dev)
    ys = torch.empty(n, dtype=tdt, device=dev)
    torch.cuda.synchronize()
    ext = torch.cuda.ExternalStream(ctx.stream, device=dev)
    for _ in range(3):
        solver.spmv(xs, ys, fused_dot=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ext)
    for _ in range(args.spmv_reps):
        solver.spmv(xs, ys, fused_dot=True)
    e1.record(ext)
    e1.synchronize()
    spmv_alone_ms = e0.elapsed_time(e1) / args.spmv_reps
    if spmv_ms is None:
        spmv_ms = spmv_alone_ms
    traffic, traffic_source = pmc_traffic("spmv_rowblock_vcp_kernel" if solver.value_codes > 0 else "spmv_rowblock_kernel")
    delta0, deltak = abs(hist[0, 0]), abs(hist[-1, 0])
    res = {
        "metric": "CG iterations/sec + SpMV effective HBM GB/s (% of 8 TB/s peak), N=10M CSR",
        "value": it_s, "unit": "CG iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"3D 7-pt Laplacian {nx}x{ny}x{nz} CSR, N={n}, nnz={nnz}, {args.dtype}, 1 RHS, "
                               f"b=5, x0=0, fixed-iteration CG ({'reference 6-op' if args.unfused else 'fused 4-launch'} loop)",
                   "rows": n, "nnz": nnz, "parallelism": "1 GPU"},
        "residual_check": {"delta_0": float(delta0), "delta_last": float(deltak), "iterations": int(hist.shape[0] - 1)},
    }
    res.update(rate_fields(spmv_moved=solver.spmv_moved_bytes, spmv_csr=spmv_bytes, spmv_ms=spmv_ms, spmv_alone_ms=spmv_alone_ms,
                           iter_moved=solver.iter_moved_bytes, iter_csr=iter_bytes, it_s=it_s, traffic=traffic,
                           traffic_source=traffic_source, n_offsets=solver.index_codes, n_values=solver.value_codes, n_pairs=solver.joint_codes))
    n_values = solver.value_codes
    solver.close()
    if n_values > 0 and not args.unfused:
        # For the record, in the same process: the same system with the value stream as passed (the HBM-bound form of the kernel; the
        # constant-coefficient stencil of this workload has 2 distinct entries, a variable-coefficient matrix runs this form).
        lib.check(lib.load().cgamd_tune(b"dev.value_codes", 0))
        try:
            s2 = pkg.Solver(ctx, n, nnz, data, indptr, indices, 1, flags=flags, dtype=dtype)
        finally:
            lib.check(lib.load().cgamd_tune(b"dev.value_codes", 1))
        s2.set_rhs(b, None, on_device=True)
        s2.iterate(args.warmup)
        ctx.synchronize()
        t0 = time.perf_counter()
        s2.iterate(args.steps)
        ctx.synchronize()
        dt2 = time.perf_counter() - t0
        ms2, _ = s2.iterate_timed(args.steps)
        moved2 = s2.spmv_moved_bytes
        res["values_as_passed"] = {"note": "same run, value codes off (dev.value_codes=0): the kernel that streams aValues",
                                   "value": args.steps / dt2, "spmv_avg_launch_ms": ms2, "moved_bytes_per_launch": moved2,
                                   "spmv_gbs": moved2 / (ms2 * 1e-3) / 1e9, "frac": moved2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "history_bit_identical": bool(np.array_equal(s2.history()[: args.warmup + args.steps + 1], hist[: args.warmup + args.steps + 1]))}
        s2.close()
    return res


if __name__ == "__main__":
    main()
