"""The LITERAL drop-in sequence of the reference's driver, on a GPU, in a fresh process without torch:

    libcg = CDLL("./build/liboclcg.so"); libcg.connect()                                   p_h-PY_C-CL.py:38-39
    libcg.cg.argtypes = [c_int, c_int, ndpointer(csingle, 1-D, C), ...]                    p_h-PY_C-CL.py:1939-1947
    libcg.cg(size, P[0].nnz, a_values, b_values, row_ptr, col_idx, x, n_my, CGMaxIT, 1)    p_h-PY_C-CL.py:1948-1950

on the `local_rect` matrix of the committed golden (the P[0] the driver passes, p_h-PY_C-CL.py:1439-1639) with n_my = 9
right-hand sides, against the oracle; then the same call at the sub-domain size the driver's docstring example gives
(~16k rows x 9 x CGMaxIT = 256) three times, recording the per-call wall split (device state / upload / setup / solve /
download) with and without the per-thread device-state cache -> gpurun_out/dropin_timing.json."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cg_numpy
import cg_oracle
from conftest import GOLDEN, PKG_NAME, ROOT

pytestmark = pytest.mark.gpu

DRIVER = r'''
import json, sys, time
import numpy as np
from ctypes import CDLL, c_int, c_double
from numpy.ctypeslib import ndpointer
assert "torch" not in sys.modules
libcg = CDLL("./build/liboclcg.so")
libcg.connect()
libcg.cg.argtypes = [c_int, c_int, ndpointer(np.csingle, ndim=1, flags="C"), ndpointer(np.csingle, ndim=1, flags="C"),
                     ndpointer(np.intc, ndim=1, flags="C"), ndpointer(np.intc, ndim=1, flags="C"),
                     ndpointer(np.csingle, ndim=1, flags="C"), c_int, c_int, c_int]
d = np.load(sys.argv[1])
out = {}
for name in ("small", "asprec"):
    row_ptr = np.array(d[name + "_indptr"], dtype=np.intc)
    col_idx = np.array(d[name + "_indices"], dtype=np.intc)
    a_values = np.array(d[name + "_data"], dtype=np.csingle)
    b_values = np.ascontiguousarray(d[name + "_b"], dtype=np.csingle)
    size, n_my = len(row_ptr) - 1, int(d[name + "_nmy"])
    calls = []
    for it in [int(v) for v in d[name + "_iters"]]:
        x = np.ascontiguousarray(np.zeros(size * n_my), dtype=np.csingle)
        t0 = time.perf_counter()
        libcg.cg(size, len(a_values), a_values, b_values, row_ptr, col_idx, x, n_my, it, 1)
        wall = (time.perf_counter() - t0) * 1e3
        split = (c_double * 6)()
        libcg.cgamd_cg_last_timing(split)
        calls.append({"iterations": it, "wall_ms": wall, "split_ms": list(split)[:5], "cache_hit": bool(split[5])})
        np.save(sys.argv[2] + "_%s_%d_%d.npy" % (name, it, len(calls)), x)
    out[name] = calls
print("DROPIN " + json.dumps(out))
'''


def _inputs(tmp_path, golden):
    g = golden["driver_generators"]
    rng = np.random.default_rng(3)
    n = len(g["local_rect_indptr"]) - 1
    n_my = 9
    b_small = np.concatenate([(r + 1) * 0.5 + rng.standard_normal(n) + 1j * rng.standard_normal(n) for r in range(n_my)])
    N = 128
    hp, hx, hd = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    b_big = np.concatenate([np.full(N * N, (r + 1) * 5.0) for r in range(n_my)])
    path = os.path.join(str(tmp_path), "in.npz")
    np.savez(path, small_indptr=g["local_rect_indptr"], small_indices=g["local_rect_indices"], small_data=g["local_rect_data"],
             small_b=b_small, small_nmy=n_my, small_iters=np.array([12, 256]),
             asprec_indptr=hp, asprec_indices=hx, asprec_data=hd, asprec_b=b_big, asprec_nmy=n_my, asprec_iters=np.array([256, 256, 256]))
    return path, (g["local_rect_indptr"], g["local_rect_indices"], g["local_rect_data"], b_small, n_my), (hp, hx, hd, b_big)


@pytest.mark.parametrize("cache", [True, False])
def test_literal_dropin_sequence(tmp_path, golden, cache):
    link = os.path.join(ROOT, "build", "liboclcg.so")
    if not os.path.exists(link):       # the path the reference's driver opens (normally made by __graft_entry__.build())
        os.makedirs(os.path.dirname(link), exist_ok=True)
        os.symlink(os.path.join("..", PKG_NAME, "liboclcg.so"), link)
    path, small, big = _inputs(tmp_path, golden)
    env = dict(os.environ, CGAMD_NO_TORCH="1")
    env["CGAMD_CG_NO_CACHE"] = "0" if cache else "1"
    prefix = os.path.join(str(tmp_path), "x")
    r = subprocess.run([sys.executable, "-c", DRIVER, path, prefix], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "error --" not in r.stderr, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("DROPIN ")][0][7:])
    # ---- parity: local_rect, 9 right-hand sides, 12 iterations (54 unknowns: later iterates are past convergence)
    ip, ix, da, b, n_my = small
    xo, _ = cg_oracle.cg(ip, ix, da.astype(np.complex128), b.astype(np.csingle).astype(np.complex128), nrhs=n_my, n_iterations=12,
                         mode=cg_oracle.MODE_SEQUENTIAL)
    x = np.load(prefix + "_small_12_1.npy")
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 2e-3
    assert np.load(prefix + "_small_256_2.npy").shape == x.shape           # CGMaxIT = 256 ran and returned
    # ---- the as_prec shape: three identical calls give identical bits (stateless), the first pays for the device state
    xs = [np.load(prefix + "_asprec_256_%d.npy" % k) for k in (1, 2, 3)]
    assert np.array_equal(xs[0], xs[1], equal_nan=True) and np.array_equal(xs[0], xs[2], equal_nan=True)
    calls = out["asprec"]
    assert [c["cache_hit"] for c in calls] == ([False, True, True] if cache else [False, False, False])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    fn = os.path.join(ROOT, "gpurun_out", "dropin_timing.json")
    prev = json.load(open(fn)) if os.path.exists(fn) else {}
    prev["cache" if cache else "no_cache"] = out
    json.dump(prev, open(fn, "w"), indent=1)


def test_stateless_cg_keeps_a_few_shapes_per_thread(pkg, gpu):
    """the reference's as_prec cycles over sub-domains of a few sizes (p_h-PY_C-CL.py:1918-1953); the stateless cg()'s
    per-thread device-state cache holds the last four shapes (one entry missed every call), an entry is reused only under the
    tuning configuration it was created with, and results do not depend on hits or misses."""
    import ctypes
    lib = pkg._lib.load()
    split = (ctypes.c_double * 6)()
    systems = []
    for N in (10, 14, 18):
        ip, ix, da = cg_numpy.poisson2d(N)
        n = N * N
        systems.append((n, ip.astype(np.intc), ix.astype(np.intc), da.astype(np.float64), np.linspace(1.0, 2.0, n)))
    lib.cgamd_cg_release_cache()

    def call(k):
        n, ip, ix, da, b = systems[k]
        x = np.zeros(n)
        pkg._lib.check(lib.cgamd_cg(pkg._lib.F64, n, len(da), pkg._lib.ptr(da), pkg._lib.ptr(b), pkg._lib.ptr(ip), pkg._lib.ptr(ix),
                                    pkg._lib.ptr(x), 1, 20, None, 0))
        lib.cgamd_cg_last_timing(split)
        return x, bool(split[5])

    first = [call(k) for k in range(3)]
    assert [h for _, h in first] == [False, False, False]
    second = [call(k) for k in range(3)]
    assert [h for _, h in second] == [True, True, True]
    for (x0, _), (x1, _) in zip(first, second):
        assert np.array_equal(x0, x1)
    pkg._lib.check(lib.cgamd_tune(b"dev.vec_ppt", 1))           # a new configuration: the cached handles were made under the old one
    try:
        x2, hit = call(0)
        assert not hit and np.allclose(x2, first[0][0], rtol=1e-12)
    finally:
        pkg._lib.check(lib.cgamd_tune(b"dev.vec_ppt", 0))
        lib.cgamd_cg_release_cache()
