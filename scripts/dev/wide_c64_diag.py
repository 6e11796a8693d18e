#!/usr/bin/env python3
"""diagnostic: chip-wide resident loop, complex64 band50000 (two-row members, 49 work-groups): run-to-run and against the launched loop"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
from test_gpu_resident import _banded_spd
from conftest import rand_vec
lib = pkg._lib.load()
ctx = pkg.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
DT = {"c64": np.complex64, "c128": np.complex128, "f64": np.float64, "f32": np.float32}[sys.argv[2] if len(sys.argv) > 2 else "c64"]
kind = sys.argv[3] if len(sys.argv) > 3 else "band"
if kind == "helm":
    import cg_numpy
    N = int(round(n ** 0.5)); n = N * N
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
elif kind == "poisson":
    import cg_numpy
    N = int(round(n ** 0.5)); n = N * N
    ip, ix, da = cg_numpy.poisson2d(N)
else:
    ip, ix, da = _banded_spd(n, 3, 0.8, 50)
rng = np.random.default_rng(n)
b = (1.0 + rand_vec(rng, n, np.complex128 if np.dtype(DT).kind == 'c' else np.float64)).astype(DT)
A = da.astype(DT)
def run(wide_min):
    pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", wide_min))
    s = pkg.Solver(ctx, n, len(ix), A, ip, ix, 1)
    s.set_rhs(b, None)
    kind = lib.cgamd_solver_loop_launches(s.handle)
    s.iterate(20)
    h = s.history().copy(); x = s.x().copy()
    s.close()
    return kind, h, x
ref = run(1 << 20)
for k in range(4):
    r = run(16)
    print("kind", r[0], "vs launched: history equal", np.array_equal(r[1], ref[1]), "max rel", float(np.max(np.abs(r[1] - ref[1]) / np.abs(ref[1]))))
pkg._lib.check(lib.cgamd_tune(b"resident_wide_min", 16))
