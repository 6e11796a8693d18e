"""Host logic of the tolerance-stopping runs on launched loops (Solver._run_to_tol; the reference's loop breaks at the first
iteration with sqrt(|r.r|) < tol, p_h-PY_C-CL.py:1338-1369, helmFE_var.py:546-586): a stand-in handle that only knows a residual
sequence checks the chunking, the shrinking check interval, the exact re-run after an overshoot, NaN and maxit.  No GPU."""
import importlib

import numpy as np
import pytest

from conftest import PKG_NAME


class FakeHandle:
    """delta_k given up front; counts what the scheme asks of the device"""

    def __init__(self, deltas):
        self.deltas = np.asarray(deltas, dtype=np.float64)
        self.done = 0
        self.set_rhs_calls = 0
        self.iterate_calls = []

    def set_rhs(self, b, x0=None):
        self.set_rhs_calls += 1
        self.done = 0

    def iterate(self, k):
        assert k >= 1
        self.iterate_calls.append(k)
        self.done += k
        assert self.done < len(self.deltas)

    def history(self):
        return self.deltas[: self.done + 1].reshape(-1, 1)

    def iterations_done(self):
        return self.done


@pytest.fixture(scope="module")
def run_to_tol():
    cl = importlib.import_module(PKG_NAME + ".cl")
    return cl.Solver._run_to_tol


def _first_below(deltas, tol):
    bad = np.nonzero(~(np.sqrt(np.abs(deltas[1:])) >= tol))[0]
    return int(bad[0]) + 1 if bad.size else None


@pytest.mark.parametrize("rate,tol,check", [(0.9, 1e-6, 8), (0.5, 1e-3, 8), (0.99, 1e-4, 16), (0.8, 1e-9, 1), (0.7, 1e-2, 5)])
def test_geometric_decay_lands_on_the_stopping_iteration(run_to_tol, rate, tol, check):
    deltas = (rate ** np.arange(4000)) ** 2 * 100.0
    want = _first_below(deltas, tol)
    h = FakeHandle(deltas)
    its = run_to_tol(h, None, None, tol, 3000, check)
    assert its == want and h.iterations_done() == want
    if want > 3 * check:
        assert h.set_rhs_calls == 1                                 # many chunks to learn the rate from: no re-run
    assert h.set_rhs_calls <= 2
    assert max(h.iterate_calls[: len(h.iterate_calls) - (1 if h.set_rhs_calls == 2 else 0)]) <= max(1, check)
    assert len(h.iterate_calls) < want / max(1, check) + 6 * max(1, int(np.log2(max(2, check)))) + 4      # not one read-back per iteration


def test_overshoot_is_re_run_exactly(run_to_tol):
    # flat, then a sudden drop in the middle of a chunk: no rate to learn from, the chunk overshoots, the solve is repeated to the exact count
    deltas = np.concatenate([np.full(21, 1.0), np.full(200, 1e-20)])
    h = FakeHandle(deltas)
    its = run_to_tol(h, None, None, 1e-6, 100, 8)
    assert its == 21 and h.iterations_done() == 21 and h.set_rhs_calls == 2 and h.iterate_calls[-1] == 21


def test_nan_stops_and_maxit_is_respected(run_to_tol):
    deltas = np.concatenate([np.linspace(4.0, 3.0, 13), [np.nan] * 50])
    h = FakeHandle(deltas)
    assert run_to_tol(h, None, None, 1e-8, 60, 8) == 13 and h.iterations_done() == 13      # first NaN: iteration 13
    deltas = np.full(500, 2.0)
    h = FakeHandle(deltas)
    assert run_to_tol(h, None, None, 1e-3, 37, 8) == 37 and h.iterations_done() == 37 and h.set_rhs_calls == 1
    h = FakeHandle(deltas)
    assert run_to_tol(h, None, None, 1e-3, 0, 8) == 0 and h.iterate_calls == []


def test_non_monotone_residual(run_to_tol):
    rng = np.random.default_rng(4)
    base = (0.93 ** np.arange(600)) ** 2
    deltas = base * np.exp(rng.uniform(-1.5, 1.5, 600))            # CG residual norms oscillate
    for tol in (1e-3, 1e-5, 1e-8):
        want = _first_below(deltas, tol)
        h = FakeHandle(deltas)
        its = run_to_tol(h, None, None, tol, 550, 8)
        assert its == want and h.iterations_done() == want and h.set_rhs_calls <= 2
