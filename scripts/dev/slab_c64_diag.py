import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, cg_numpy, cg_oracle
pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
dmod = importlib.import_module("conjugate-gradient-pyopencl_amd.dist")
ctx = pkg.Context(0); dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dtype = np.complex64
ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
b = np.tile(cg_numpy.rhsA(100, 12.0).flatten(), (N * N) // 10000)
n = len(ip) - 1
plan = dmod.build_halo_plan(torch.from_numpy(ix.astype(np.int64)).to(dev), [(0, n)], 0)
vals = torch.from_numpy(da.astype(dtype)).to(dev); indptr = torch.from_numpy(ip.astype(np.int32)).to(dev)
bl = torch.from_numpy(b.astype(dtype)).to(dev)
res = {}
for name, fl in (("slab", 512), ("launched", 0)):
    s = dmod.DistSolver(ctx, plan, indptr, vals, dtype, flags=fl)
    s.set_rhs(bl, None); s.iterate(36)
    res[name] = (s.x(torch.empty(n, dtype=torch.complex64, device=dev)).cpu().numpy(), s.history().copy())
    print(name, "launches", pkg._lib.load().cgamd_dist_loop_launches(s.handle))
    s.close()
xo, ho = cg_oracle.cg(ip, ix, da.astype(np.complex128), b.astype(np.complex128), n_iterations=36, mode=cg_oracle.MODE_FAST)
for name in res:
    h = res[name][1]
    dev_ = np.abs(h - ho[:, 0]) / np.abs(ho[:, 0])
    print(name, "max dev k<=12 %.2e  k<=24 %.2e  k<=36 %.2e" % (dev_[:13].max(), dev_[:25].max(), dev_.max()), " x err %.2e" % (np.linalg.norm(res[name][0] - xo) / np.linalg.norm(xo)))
d2 = np.abs(res["slab"][1] - res["launched"][1]) / np.abs(res["launched"][1])
print("slab vs launched: k<=12 %.2e k<=24 %.2e k<=36 %.2e" % (d2[:13].max(), d2[:25].max(), d2.max()))
print("delta", np.abs(ho[::6, 0]))
