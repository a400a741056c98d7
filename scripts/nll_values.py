#!/usr/bin/env python3
"""Developer: the objective of configs[1]'s tiles at fixed parameters over REPS launches -- how many distinct values does a
tile take (GPSAT_LIB selects the build)?"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

T, N, P, D, kid = int(os.environ.get("GD_T", "4096")), int(os.environ.get("GD_N", "500")), 8, 3, 0
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
WG = os.environ.get("GD_GRAD", "1") == "1"
with threadpool_limits(1):
    with ThreadPoolExecutor(16) as pool:
        tiles = list(pool.map(lambda t: syn.make_tile(t, N, P, D, kid), range(T)))
X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
th = np.exp(np.random.default_rng(5).normal(0.0, 0.5, (T, D + 2)))
kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
          theta0=th, kernel="RBF", optimiser="none", want_grad=WG)
eng = Engine(0, workgroups_per_cu=int(os.environ.get("GD_WG", "0")))       # GD_WG=1: the 8-wave build
nl, gdiff, g0 = [], np.zeros(T, dtype=np.int64), None
for _ in range(REPS):
    r = eng.fit_predict_batch(X=X, y=y, Xs=Xs, **kw)
    nl.append(r.nll.copy())
    if WG:
        if g0 is None:
            g0 = r.grad.copy()
        gdiff += (r.grad != g0).any(axis=1)
nl = np.stack(nl)
eng.close()
if WG:
    print("tiles whose gradient differed from the first launch's at least once:", int((gdiff > 0).sum()), "of", T, "(N =", N, ")")
ndist = np.array([len(np.unique(nl[:, t])) for t in range(T)])
print(os.path.basename(os.environ.get("GPSAT_LIB", "default")), "want_grad", WG, "tiles with 1 / 2 / 3 / more distinct objective values over", REPS, "launches:",
      [(ndist == 1).sum(), (ndist == 2).sum(), (ndist == 3).sum(), (ndist > 3).sum()])
for t in np.nonzero(ndist > 1)[0][:6]:
    u, c = np.unique(nl[:, t], return_counts=True)
    print("  tile", t, "values", [f"{v:.10f}" for v in u], "counts", c.tolist(), "spread / |nll|", float((u.max() - u.min()) / abs(u[0])))
