"""Several PROCESSES on one GPU, each calling the literal drop-in cg() concurrently -- what the reference's MPI driver does
when ranks share a device (p_h-PY_C-CL.py: every rank's as_prec calls libcg.cg).  Small systems run the resident loop, whose
groups of work-groups must all be running at once; launches are therefore serialised per GPU across processes (an advisory
file lock, csrc/resident.hip).  Every process must get the bits a lone process gets."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cg_numpy
from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import sys, time
import numpy as np
from ctypes import CDLL, c_int
from numpy.ctypeslib import ndpointer
libcg = CDLL("./build/liboclcg.so")
libcg.connect()
libcg.cg.argtypes = [c_int, c_int, ndpointer(np.csingle, ndim=1, flags="C"), ndpointer(np.csingle, ndim=1, flags="C"),
                     ndpointer(np.intc, ndim=1, flags="C"), ndpointer(np.intc, ndim=1, flags="C"),
                     ndpointer(np.csingle, ndim=1, flags="C"), c_int, c_int, c_int]
d = np.load(sys.argv[1])
row_ptr, col_idx, a_values, b = d["indptr"], d["indices"], d["data"], d["b"]
size, n_my, calls = len(row_ptr) - 1, int(d["nmy"]), int(sys.argv[3])
# start line: all workers begin their calls together
while time.time() < float(sys.argv[4]):
    time.sleep(0.001)
xs = []
for c in range(calls):
    x = np.zeros(size * n_my, dtype=np.csingle)
    libcg.cg(size, len(a_values), a_values, b, row_ptr, col_idx, x, n_my, 64, 1)
    xs.append(x)
assert all(np.array_equal(xs[0], x) for x in xs[1:])
np.save(sys.argv[2], xs[0])
print("WORKER done")
'''


@pytest.mark.timeout(300)
def test_three_processes_share_one_gpu_through_the_drop_in(tmp_path):
    import time
    N, n_my = 128, 9
    hp, hx, hd = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    rng = np.random.default_rng(7)
    b = (rng.standard_normal(N * N * n_my) + 1j * rng.standard_normal(N * N * n_my)).astype(np.csingle)
    inp = str(tmp_path / "in.npz")
    np.savez(inp, indptr=hp.astype(np.intc), indices=hx.astype(np.intc), data=hd.astype(np.csingle), b=b, nmy=n_my)
    script = str(tmp_path / "worker.py")
    with open(script, "w") as f:
        f.write(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def launch(tag, calls, start):
        return subprocess.Popen([sys.executable, script, inp, str(tmp_path / f"x_{tag}.npy"), str(calls), str(start)], cwd=ROOT, env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    p = launch("alone", 1, 0.0)
    out, _ = p.communicate(timeout=120)
    assert p.returncode == 0 and "WORKER done" in out, out[-2000:]
    start = time.time() + 25.0           # leaves the three interpreters time to load the library and the HIP runtime
    procs = [launch(f"p{i}", 40, start) for i in range(3)]
    outs = [q.communicate(timeout=240)[0] for q in procs]
    for q, o in zip(procs, outs):
        assert q.returncode == 0 and "WORKER done" in o, o[-2000:]
        assert "error --" not in o, o[-2000:]
    ref = np.load(tmp_path / "x_alone.npy")
    assert np.all(np.isfinite(ref))
    for i in range(3):
        assert np.array_equal(np.load(tmp_path / f"x_p{i}.npy"), ref)
