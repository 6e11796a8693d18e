"""How far is the fp32 / complex64 GPU build from the reference OpenCL solver at the reference's OWN precision?

The reference's kernels cannot be executed here (no OpenCL device, no pyopencl: SURVEY 8c), so "parity to OpenCL arithmetic"
is restated, not pinned: the C oracle runs the reference's op structure in fp32 / complex64 in the reference's summation
order (32-lane strided + tree SpMV rows, 256-wide tree + sequential host sum for the dots,
kernel/real/vdot.cl:2-38, clcg.c:274-279) -- MODE_REFERENCE_ORDER -- and the device results are compared with it, next to
the fp64 oracle.  Three distances per system (max over iterations k with delta_k/delta_0 > 1e-4, k <= K):
    a = |gpu32 - ref32| / |ref64|     b = |gpu32 - ref64| / |ref64|     c = |ref32 - ref64| / |ref64|
The build accumulates its dot products in fp64 and sums SpMV rows in CSR order, the reference sums in fp32 trees: the two
differ from each other by fp32 rounding amplified by the recurrence, i.e. by about c.  Asserted: a <= 3 c + 1e-5 (the
build sits inside the rounding ball of the reference-precision arithmetic) and b <= 3 c + 1e-5 (it is not further from the
fp64 truth than a small multiple of what the reference-precision arithmetic is; at N = 1M it is 4000x closer, because the
reference adds 3907 work-group partials sequentially in fp32).  The measured values go to gpurun_out/refprec.json (copied to
profiles/r2/refprec.json, quoted in DESIGN.md section 2)."""
import json
import os

import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

RESULTS = {}


def _distances(pkg, gpu, name, ip, ix, da64, b64, dtype, iters, K):
    ctx, queue, kernels = gpu
    wide = np.complex128 if np.dtype(dtype).kind == "c" else np.float64
    n = len(ip) - 1
    x32 = np.zeros(n, dtype=dtype)
    x32, h32 = pkg.CG(ctx, queue, kernels, n, len(ix), da64.astype(dtype), b64.astype(dtype), ip, ix, x32, 1, iters, return_history=True)
    xr, hr = cg_oracle.cg(ip, ix, da64.astype(dtype), b64.astype(dtype), n_iterations=iters, mode=cg_oracle.MODE_REFERENCE_ORDER, dtype=dtype)
    xw, hw = cg_oracle.cg(ip, ix, da64.astype(dtype).astype(wide), b64.astype(dtype).astype(wide), n_iterations=iters, mode=cg_oracle.MODE_SEQUENTIAL)
    keep = np.abs(hw[:, 0]) / np.abs(hw[0, 0]) > 1e-4
    keep[K + 1:] = False
    den = np.abs(hw[keep, 0])
    a = float(np.max(np.abs(h32[keep, 0] - hr[keep, 0]) / den))
    b = float(np.max(np.abs(h32[keep, 0] - hw[keep, 0]) / den))
    c = float(np.max(np.abs(hr[keep, 0] - hw[keep, 0]) / den))
    nx = np.linalg.norm(xw)
    res = {"system": name, "dtype": np.dtype(dtype).name, "rows": int(n), "iterations": int(iters), "compared_k": int(keep.sum() - 1),
           "delta_gpu_vs_ref32": a, "delta_gpu_vs_fp64": b, "delta_ref32_vs_fp64": c,
           "x_gpu_vs_ref32": float(np.linalg.norm(x32 - xr) / nx), "x_gpu_vs_fp64": float(np.linalg.norm(x32 - xw) / nx),
           "x_ref32_vs_fp64": float(np.linalg.norm(xr - xw) / nx)}
    RESULTS[name + "/" + np.dtype(dtype).name] = res
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(RESULTS, open(os.path.join(ROOT, "gpurun_out", "refprec.json"), "w"), indent=1)
    return res


def _check(res):
    a, b, c = res["delta_gpu_vs_ref32"], res["delta_gpu_vs_fp64"], res["delta_ref32_vs_fp64"]
    assert a <= 3.0 * c + 1e-5, res
    assert b <= 3.0 * c + 1e-5, res
    assert res["x_gpu_vs_ref32"] <= 3.0 * res["x_ref32_vs_fp64"] + 1e-5, res


def test_helm32_complex64_vs_reference_precision_oracle(pkg, gpu, golden):
    g = golden["cg_iterates"]
    _check(_distances(pkg, gpu, "helmFE_var(32)", g["helm32_indptr"], g["helm32_indices"], g["helm32_data"], g["helm32_b"],
                      np.complex64, 50, 40))


def test_poisson40_float32_vs_reference_precision_oracle(pkg, gpu):
    ip, ix, da = cg_numpy.poisson2d(40)
    _check(_distances(pkg, gpu, "Poisson(40)", ip, ix, da, np.linspace(1.0, 2.0, 1600), np.float32, 50, 40))


def test_config3_full_size_complex64_vs_reference_precision_oracle(pkg, gpu):
    """BASELINE config 3 at full size in the reference's own type: helmFE_var(N=500), n = 250 000, complex64"""
    N = 500
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    b = cg_numpy.rhsA(N, 12.0).flatten()
    cg_oracle.set_threads(16)
    _check(_distances(pkg, gpu, "helmFE_var(500) [C3]", ip, ix, da, b, np.complex64, 60, 50))


def test_config2_full_size_float32_vs_reference_precision_oracle(pkg, gpu):
    """the config-2 matrix (Poisson(1000), n = 1M) in fp32 -- the reference's real type"""
    ip, ix, da = cg_numpy.poisson2d(1000)
    b = np.full(1000 * 1000, 5.0)
    cg_oracle.set_threads(16)
    _check(_distances(pkg, gpu, "Poisson(1000) [C2 matrix]", ip, ix, da, b, np.float32, 40, 30))
