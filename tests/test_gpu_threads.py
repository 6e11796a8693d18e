"""GPU tests of the reference's threaded multi-GPU mode (p_h-PY_C-CL-multi-GPU.py:2123-2181): the right-hand sides are
sharded over devices, ONE PYTHON THREAD PER DEVICE, each with its own context / queue / kernels, matrix replicated, no
communication; slices are copied back under a lock.  One GPU box: the same physical device is named several times, so
the threads really run concurrently on one GPU -- what has to hold is thread safety per handle (SURVEY 8b "Threading")."""
import threading

import numpy as np
import pytest

import cg_numpy
from conftest import rand_vec

pytestmark = pytest.mark.gpu


def _asprec_system(N=64, n_rhs=9):
    ip, ix, da = cg_numpy.helm_fe_var(N, 12.0, np.ones((N - 1, N - 1)), 0.15, N, N)
    n = N * N
    rng = np.random.default_rng(7)
    b = np.concatenate([(r + 1) * 5.0 + rand_vec(rng, n, np.complex64) for r in range(n_rhs)]).astype(np.csingle)
    return n, ip.astype(np.intc), ix.astype(np.intc), da.astype(np.csingle), b


@pytest.mark.parametrize("n_threads", [3, 4])
def test_rhs_sharded_threads_match_the_serial_solve_bit_for_bit(pkg, gpu, n_threads):
    ctx, queue, kernels = gpu
    n_rhs, iters = 9, 64
    n, ip, ix, da, b = _asprec_system(64, n_rhs)
    x_serial = np.zeros(n * n_rhs, dtype=np.csingle)
    pkg.CG(ctx, queue, kernels, n, len(da), da, b, ip, ix, x_serial, n_rhs, iters)
    assert np.all(np.isfinite(x_serial))
    devices = [pkg.Device(0) for _ in range(n_threads)]           # one entry = one context + stream + thread
    workloads = pkg.distribute_workloads_on_devices(devices, n_rhs)
    spans = sorted((w[0], w[1]) for w in workloads.values())
    assert spans[0][0] == 0 and spans[-1][1] == n_rhs and all(a[1] == b_[0] for a, b_ in zip(spans, spans[1:]))
    assert sorted((e - s for s, e in spans), reverse=True) == sorted(
        [n_rhs // n_threads + (1 if i < n_rhs % n_threads else 0) for i in range(n_threads)], reverse=True)
    assert len({id(w[2]) for w in workloads.values()}) == n_threads
    for rep in range(3):                                             # repeated: races would not show every time
        x = np.zeros(n * n_rhs, dtype=np.csingle)
        out = pkg.distribute_computations_with_threads(n, len(da), da, b, ip, ix, x, n_rhs, iters, workloads)
        assert out is x
        assert np.array_equal(x, x_serial), rep
    for w in workloads.values():
        w[2].close()
    x = np.zeros(n * n_rhs, dtype=np.csingle)
    pkg.solve_rhs_sharded([pkg.Device(0) for _ in range(n_threads)], n, len(da), da, b, ip, ix, x, n_rhs, iters)
    assert np.array_equal(x, x_serial)


def test_handles_keep_their_configuration_while_other_threads_tune(pkg, gpu):
    """every solver runs with the configuration snapshot it was created under: a thread that flips tuning knobs in a loop
    must not change what concurrently running solvers compute (they were created before, or create their own snapshot)"""
    ctx0, queue, kernels = gpu
    lib = pkg._lib.load()
    n_rhs, iters = 3, 48
    n, ip, ix, da, b = _asprec_system(48, n_rhs)
    ref = {}
    for dt in (np.complex64, np.complex128):
        x = np.zeros(n * n_rhs, dtype=dt)
        ref[dt] = pkg.CG(ctx0, queue, kernels, n, len(da), da.astype(dt), b.astype(dt), ip, ix, x, n_rhs, iters).copy()
    stop, errors, results = threading.Event(), [], {}

    def tuner():
        i = 0
        while not stop.is_set():
            for key, vals in ((b"vec_grid", (0, 64, 256)), (b"spmv_cycle", (64, 8)), (b"two_launch", (1, 0)), (b"dev.no_fold_alpha", (0, 1))):
                lib.cgamd_tune(key, vals[i % len(vals)])
            i += 1

    def worker(k, dt):
        try:
            c = pkg.Context(0)
            s = pkg.Solver(c, n, len(da), da.astype(dt), ip, ix, n_rhs)       # snapshot taken here, whatever it is
            outs = []
            for _ in range(4):
                s.set_rhs(b.astype(dt), None)
                s.iterate(iters)
                outs.append(s.x())
            s.close()
            c.close()
            results[k] = (dt, outs)
        except BaseException as e:      # noqa: BLE001
            errors.append(e)

    tt = threading.Thread(target=tuner)
    tt.start()
    workers = [threading.Thread(target=worker, args=(k, (np.complex64, np.complex128)[k % 2])) for k in range(4)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    stop.set()
    tt.join()
    for key in (b"vec_grid", b"spmv_cycle", b"two_launch", b"dev.no_fold_alpha"):
        lib.cgamd_tune(key, {b"vec_grid": 0, b"spmv_cycle": 64, b"two_launch": 1}.get(key, 0))
    assert not errors, errors
    for k, (dt, outs) in results.items():
        # one handle = one configuration: its four solves agree bit for bit ...
        assert all(np.array_equal(o, outs[0]) for o in outs[1:]), k
        # ... and whatever loop variant the snapshot selected computes the same recurrence as the default one
        tol = 1e-9 if dt == np.complex128 else 2e-3
        assert np.linalg.norm(outs[0] - ref[dt]) / np.linalg.norm(ref[dt]) < tol, k


def test_multi_gpu_worker_honours_its_device_argument(pkg, gpu):
    ctx, queue, kernels = gpu
    n, ip, ix, da, b = _asprec_system(32, 2)
    x = np.zeros(n * 2, dtype=np.csingle)
    out = pkg.conjugate_gradient_multi_gpu(ctx, queue, kernels, n, len(da), da, b, ip, ix, x, 2, 10, pkg.Device(0))
    assert out is x and np.all(np.isfinite(x))
    x2 = np.zeros(n * 2, dtype=np.csingle)
    pkg.conjugate_gradient_multi_gpu(ctx, queue, kernels, n, len(da), da, b, ip, ix, x2, 2, 10, 0)     # plain index
    assert np.array_equal(x, x2)
    with pytest.raises(pkg.CgAmdError):                        # a device that does not exist: loud failure, not device 0
        pkg.conjugate_gradient_multi_gpu(ctx, queue, kernels, n, len(da), da, b, ip, ix, x2, 2, 10, len(pkg.get_gpu_devices()) + 3)
