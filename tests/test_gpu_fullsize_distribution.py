"""GPU: parity DISTRIBUTION at BASELINE size -- configs[1]'s 4096 distinct tiles (N = 500, P = 500,
RBF, D = 3) through (a) the fp32 kernels with the bench's settings (max_iter 20, default tolerances), (b) the fp32 kernels
run to convergence and (c) the fp64 kernels run to convergence (pinned to the oracle at 1e-9 / to SciPy's optimum at the
reference's own 1e-3 / 1e-6 by tests/test_gpu_sklearn_pins.py), and asserts quantiles over ALL tiles.

Stated bounds (measured values in brackets: scripts/parity_distribution.py, round 3):
                                   (a) vs (c)                          (b) vs (c)
  |dl| / l              median 2e-3 [6.0e-4]  p99 8e-2 [3.5e-2]   median 2e-3 [5.4e-4]  p99 7e-2 [2.9e-2]
  |d sf2| / sf2         median 4e-3 [1.3e-3]  p99 0.15 [6.7e-2]   median 4e-3 [1.2e-3]  p99 0.15 [6.5e-2]
  |d sn2| / sn2         median 5e-4 [1.5e-4]  p99 1e-2 [3.4e-3]   median 5e-4 [1.4e-4]  p99 8e-3 [2.4e-3]
  (NLL - NLL_c) / N     median 5e-7 [1.3e-7]  p99 2e-4 [5.6e-5]   median 5e-7 [1.1e-7]  p99 1.2e-4 [3.4e-5]
                        max 3e-3 [8.4e-4], min -2e-5 [-3.5e-6]
  max|df*| / max|y|     median 4e-4 [1.0e-4]  p99 8e-3 [2.5e-3]   median 4e-4 [9.5e-5]  p99 7e-3 [2.0e-3]
  max|df*_var| / sf2    median 5e-5 [1.0e-5]  p99 2e-3 [4.9e-4]   median 5e-5 [9.3e-6]  p99 1.5e-3 [3.6e-4]
Length scales and the kernel variance are weakly determined along the ridge l -> bound (10 % of the tiles end with a length
scale at the upper bound of its box): the tails of their relative differences are wide while the objective and the
predictions agree tightly -- which is why the objective gap per observation is the headline quality figure
(bench.py `quality`).  The reference's own CPU-vs-GPU tolerance (tests/integration.py:109-132: 1e-3 length scales, 1e-6
variances) is an fp64-vs-fp64 figure; tests/test_gpu_sklearn_pins.py holds the fp64 kernels to it."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from gpsat_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _q(v):
    v = np.asarray(v, dtype=np.float64)
    assert np.isfinite(v).all()
    return float(np.median(v)), float(np.quantile(v, 0.99)), float(v.max()), float(v.min())


def test_full_size_distribution_fp32_vs_fp64_converged():
    from threadpoolctl import threadpool_limits
    from gpsat_amd.engine import Engine
    T, N, P, D, kid = 4096, 500, 500, 3, 0
    with threadpool_limits(1):
        with ThreadPoolExecutor(16) as pool:                  # threads, not processes: this process already holds the GPU
            tiles = list(pool.map(lambda t: syn.make_tile(1_000_000 * 0 + t, N, P, D, kid), range(T)))
    X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
    y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
    Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
    del tiles
    lo, hi = syn.default_bounds(T, D)
    kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
              theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs")
    eng = Engine(0)
    try:
        r_a = eng.fit_predict_batch(X=X, y=y, Xs=Xs, max_iter=20, **kw)
        r_b = eng.fit_predict_batch(X=X, y=y, Xs=Xs, max_iter=500, **kw)
        r_c = eng.fit_predict_batch(X=X.astype(np.float64), y=y.astype(np.float64), Xs=Xs.astype(np.float64), max_iter=500,
                                    dtype="f64", **kw)
    finally:
        eng.close()
    assert (r_c.status == 0).all() and (r_b.status == 0).mean() > 0.999 and np.isin(r_a.status, (0, 1)).all()
    ymax = np.abs(y.astype(np.float64)).reshape(T, N).max(axis=1)
    bounds = {  # (median, p99) for (a) and (b)
        "l": ((2e-3, 8e-2), (2e-3, 7e-2)), "sf2": ((4e-3, 0.15), (4e-3, 0.15)), "sn2": ((5e-4, 1e-2), (5e-4, 8e-3)),
        "nll": ((5e-7, 2e-4), (5e-7, 1.2e-4)), "f": ((4e-4, 8e-3), (4e-4, 7e-3)), "fv": ((5e-5, 2e-3), (5e-5, 1.5e-3))}
    for which, r in enumerate((r_a, r_b)):
        d = {"l": np.abs(r.theta[:, :D] - r_c.theta[:, :D]) / r_c.theta[:, :D],
             "sf2": np.abs(r.theta[:, D] - r_c.theta[:, D]) / r_c.theta[:, D],
             "sn2": np.abs(r.theta[:, D + 1] - r_c.theta[:, D + 1]) / r_c.theta[:, D + 1],
             "nll": (r.nll - r_c.nll) / N,
             "f": np.max(np.abs(r.f_mean.astype(np.float64).reshape(T, P) - r_c.f_mean.reshape(T, P)), axis=1) / ymax,
             "fv": np.max(np.abs(r.f_var.astype(np.float64).reshape(T, P) - r_c.f_var.reshape(T, P)), axis=1) / r_c.theta[:, D]}
        for k, v in d.items():
            med, p99, mx, mn = _q(v)
            bm, b99 = bounds[k][which]
            assert med <= bm and p99 <= b99, (("bench settings", "converged")[which], k, med, p99)
            if k == "nll":
                assert mx <= 3e-3 and mn >= -2e-5, (which, mx, mn)
    # the early stop of the bench's settings costs almost nothing against running fp32 to convergence
    assert abs(float(r_a.n_eval.mean()) - float(r_b.n_eval.mean())) < 1.5
    # ---- an ORACLE value enters the full-size check: 64 of the 4096 tiles, objective and predictions of
    # the fp64 oracle AT THE PARAMETERS THE GPU RETURNED, against the bench-settings fp32 run (stated fp64 -> fp32 bounds of
    # tests/test_gpu_parity.py) and the fp64 run (1e-9 relative)
    from oracle import gp_oracle as go
    with threadpool_limits(4):
        for t in range(17, T, 64):
            Xt, yt, Xst = (X[t * N:(t + 1) * N].astype(np.float64), y[t * N:(t + 1) * N].astype(np.float64),
                           Xs[t * P:(t + 1) * P].astype(np.float64))
            for r, tol_nll, tol_f, tol_v in ((r_a, 2e-5 * N, 2e-3, 2e-3), (r_c, 1e-9, 1e-8, 1e-8)):
                f, _ = go.nll_and_grad(kid, Xt, yt, r.theta[t], want_grad=False)
                fm, fv, _ = go.predict(kid, Xt, yt, Xst, r.theta[t])
                scale_nll = 1.0 if r is r_a else abs(f)
                assert abs(r.nll[t] - f) <= tol_nll * scale_nll + 2e-6 * abs(f) * (r is r_a), (t, r.nll[t], f)
                assert np.max(np.abs(np.asarray(r.f_mean[t * P:(t + 1) * P], dtype=np.float64) - fm)) <= tol_f * ymax[t]
                assert np.max(np.abs(np.asarray(r.f_var[t * P:(t + 1) * P], dtype=np.float64) - fv)) <= tol_v * r.theta[t, D] + 1e-6 * (r is r_a)
