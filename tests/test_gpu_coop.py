"""GPU: cooperative tiles (gpsat_amd/csrc/gpsat_coop.h) -- a workgroup without a tile of its own helps a running tile by
pulling groups of its sweep and gradient queues.  Helped or not, every output is the same bits: per
group the arithmetic is the same whoever runs it, per-column updates are ordered by flags, the gradient's partial sums are
added in a fixed order.  GPSAT_DEBUG_COOP: 0 = off, 1 = on (default), 2 = the cooperative code path in every evaluation of
a helpable tile, helped or not."""
import os

import numpy as np
import pytest

from gpsat_amd import synthetic as syn

pytestmark = pytest.mark.gpu
os.environ["GPSAT_DEVELOPER"] = "1"       # the GPSAT_DEBUG_* knobs used below are read in developer mode only


@pytest.fixture(scope="module")
def eng():
    from gpsat_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    os.environ.pop("GPSAT_DEBUG_COOP", None)
    os.environ.pop("GPSAT_DEBUG_GRID", None)


def _run(eng, b, kernel, mode, grid=None, **kw):
    os.environ["GPSAT_DEBUG_COOP"] = str(mode)
    if grid is not None:
        os.environ["GPSAT_DEBUG_GRID"] = str(grid)
    try:
        T, D = b["T"], b["D"]
        lo, hi = syn.default_bounds(T, D)
        return eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                                     theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel=kernel, **kw)
    finally:
        os.environ.pop("GPSAT_DEBUG_COOP", None)
        os.environ.pop("GPSAT_DEBUG_GRID", None)


def _same(a, b):
    for f in ("theta", "nll", "status", "n_eval", "n_iter", "f_mean", "f_var", "y_var"):
        np.testing.assert_array_equal(np.asarray(getattr(a, f)), np.asarray(getattr(b, f)), err_msg=f)
    if a.grad is not None:
        np.testing.assert_array_equal(a.grad, b.grad)


@pytest.mark.parametrize("kernel,Ns", [
    ("Matern32", [2048]), ("RBF", [1024]), ("Matern52", [1200]), ("Matern12", [900]), ("RBF", [500]), ("Matern32", [416]),
    ("Matern32", [2048, 1536, 1024, 1024, 768, 640, 512, 512, 500, 400, 384, 300, 256, 200, 128, 100, 64, 33, 32, 31, 1, 0, 700, 900, 1200]),
])
def test_helped_tiles_are_bit_identical(eng, kernel, Ns):
    b = syn.make_batch(len(Ns), Ns, 37, 3, {"RBF": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3}[kernel], base_seed=77)
    kw = dict(optimiser="lbfgs", max_iter=5, want_grad=True)
    off = _run(eng, b, kernel, 0, **kw)
    on = _run(eng, b, kernel, 1, **kw)
    forced = _run(eng, b, kernel, 2, **kw)
    alone = _run(eng, b, kernel, 2, grid=len(Ns), **kw)          # cooperative code path, no spare workgroup in the launch
    _same(off, on)
    _same(off, forced)
    _same(off, alone)
    assert np.isin(off.status[np.array(Ns) > 0], (0, 1)).all()


def test_helpers_shorten_a_large_tile(eng):
    b = syn.make_batch(1, [2048], 16, 3, 2, base_seed=5)
    kw = dict(optimiser="lbfgs", max_iter=6)
    _run(eng, b, "Matern32", 1, **kw)
    t_off = min(_run(eng, b, "Matern32", 0, **kw).kernel_ms for _ in range(2))
    t_on = min(_run(eng, b, "Matern32", 1, **kw).kernel_ms for _ in range(2))
    assert t_on < 0.75 * t_off, (t_on, t_off)


def test_not_positive_definite_in_a_helped_tile_is_reported(eng):
    """A failing evaluation inside a cooperative phase unwinds owner and helpers (bounded waits, no hang) and is reported
    as for an unhelped tile."""
    N = 1024
    X = np.zeros((N, 1), dtype=np.float32)
    X[:, 0] = np.repeat(np.arange(N // 2), 2)                     # duplicate points, (near) zero noise: not PD in fp32
    y = np.ones(N, dtype=np.float32)
    for mode in (0, 2):
        os.environ["GPSAT_DEBUG_COOP"] = str(mode)
        try:
            r = eng.fit_predict_batch(D=1, obs_off=[0, N], X=X, y=y, pred_off=[0, 2], Xs=np.zeros((2, 1), np.float32),
                                      theta0=[[1.0, 1.0, 1e-12]], kernel="RBF", optimiser="none")
        finally:
            os.environ.pop("GPSAT_DEBUG_COOP", None)
        assert r.status[0] == 2 and np.isnan(r.nll[0]) and np.isnan(r.f_mean).all()
