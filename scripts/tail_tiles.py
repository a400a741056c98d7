"""Developer: which of configs[1]'s 4096 tiles take the most evaluations, and what do the extra evaluations buy?  (E46: the
end-of-launch idle time of the headline run is these tiles.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from threadpoolctl import threadpool_limits
from gpsat_amd import synthetic as syn
from gpsat_amd.engine import Engine

T, N, P, D, kid = 4096, 500, 8, 3, 0
with threadpool_limits(1):
    with ThreadPoolExecutor(16) as pool:
        tiles = list(pool.map(lambda t: syn.make_tile(t, N, P, D, kid), range(T)))
X = np.concatenate([t[0] for t in tiles]).astype(np.float32); y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
lo, hi = syn.default_bounds(T, D)
kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P, theta0=np.ones((T, D + 2)), lo=lo, hi=hi,
          kernel="RBF", optimiser="lbfgs")
eng = Engine(0)
r = eng.fit_predict_batch(X=X, y=y, Xs=Xs, max_iter=20, **kw)
rc = eng.fit_predict_batch(X=X.astype(np.float64), y=y.astype(np.float64), Xs=Xs.astype(np.float64), max_iter=500, dtype="f64", **kw)
gap = (r.nll - rc.nll) / N
print("evals/tile", r.n_eval.mean(), "iters", r.n_iter.mean(), "status counts", np.bincount(r.status, minlength=7))
print("n_eval histogram (bins of 5):", np.bincount(np.minimum(r.n_eval, 70) // 5))
for lo_, hi_ in ((0, 20), (20, 30), (30, 40), (40, 50), (50, 100)):
    m = (r.n_eval >= lo_) & (r.n_eval < hi_)
    if m.any():
        print(f"n_eval in [{lo_},{hi_}): {m.sum()} tiles, iters {r.n_iter[m].mean():.1f}, evals/iter {np.mean(r.n_eval[m] / np.maximum(r.n_iter[m], 1)):.2f}, status {np.bincount(r.status[m], minlength=7).tolist()},"
              f" nll gap/obs median {np.median(gap[m]):.2e} p90 {np.quantile(gap[m], 0.9):.2e}")
for ml in (3, 5, 8):
    r2 = eng.fit_predict_batch(X=X, y=y, Xs=Xs, max_iter=20, max_ls=ml, **kw)
    g2 = (r2.nll - rc.nll) / N
    print(f"max_ls={ml}: evals/tile {r2.n_eval.mean():.2f} max {r2.n_eval.max()} status {np.bincount(r2.status, minlength=7).tolist()} gap median {np.median(g2):.2e} p99 {np.quantile(g2, 0.99):.2e} max {g2.max():.2e} kernel_ms {r2.kernel_ms:.1f}")
print("default kernel_ms", r.kernel_ms, "gap median", np.median(gap), "p99", np.quantile(gap, 0.99), "max", gap.max())
