#!/usr/bin/env python3
"""Developer: status histogram / evaluation counts of BASELINE configs[1]'s 4096 tiles in fp32 (max_iter 20 and run to
convergence) for the library build GPSAT_LIB selects -- the data of tests/test_gpu_fullsize_distribution.py."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

T, N, P, D, kid = 4096, 500, 500, 3, 0
with threadpool_limits(1):
    with ThreadPoolExecutor(16) as pool:
        tiles = list(pool.map(lambda t: syn.make_tile(t, N, P, D, kid), range(T)))
X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
lo, hi = syn.default_bounds(T, D)
kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
          theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs")
eng = Engine(0)
tag = os.path.basename(os.environ.get("GPSAT_LIB", "default"))
for mi in (20, 500):
    r = eng.fit_predict_batch(X=X, y=y, Xs=Xs, max_iter=mi, **kw)
    u, c = np.unique(r.status, return_counts=True)
    print(tag, "max_iter", mi, "status", dict(zip(u.tolist(), c.tolist())), "evals", round(float(r.n_eval.mean()), 2),
          "iters", round(float(r.n_iter.mean()), 2), "odd tiles", np.nonzero(~np.isin(r.status, (0, 1)))[0][:8].tolist(), flush=True)
eng.close()
