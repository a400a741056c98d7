#!/usr/bin/env python3
"""Developer: the same 100 000-tile batch (N = 500, the configs[3] test's data) run REPS times; every run is compared with the
first, mismatching tiles are counted and described.  GPSAT_LIB / GPSAT_DEBUG_* select the build and the scheduling."""
import os
os.environ.setdefault("GPSAT_DEVELOPER", "1")     # GPSAT_DEBUG_* knobs are read in developer mode only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402

REPS = int(os.environ.get("REPS", 4))
T, N, P, D, kid, NPROTO = int(os.environ.get("T", 100_000)), 500, 20, 3, 0, 500
proto = [syn.make_tile(31_000 + j, N, P, D, kid) for j in range(NPROTO)]
scale = 1.0 + 0.01 * (np.arange(T) // NPROTO)
X = np.tile(np.concatenate([p[0] for p in proto]).astype(np.float32), (T // NPROTO, 1))
Xs = np.tile(np.concatenate([p[2] for p in proto]).astype(np.float32), (T // NPROTO, 1))
y = (np.tile(np.concatenate([p[1] for p in proto]), T // NPROTO) * np.repeat(scale, N)).astype(np.float32)
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, X=X, y=y, pred_off=np.arange(T + 1, dtype=np.int64) * P, Xs=Xs,
          theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", max_iter=20)
ref = eng.fit_predict_batch(**kw)
tag = os.path.basename(os.environ.get("GPSAT_LIB", "default")) + " " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("GPSAT_DEBUG"))
for rep in range(REPS):
    r = eng.fit_predict_batch(**kw)
    bad = np.nonzero((r.theta != ref.theta).any(axis=1) | (r.nll != ref.nll) | (r.n_eval != ref.n_eval))[0]
    print(tag, "run", rep, "kernel_ms", round(r.kernel_ms, 1), "mismatching tiles", len(bad), flush=True)
    for t in bad[:4]:
        print("    tile", int(t), "n_eval", int(ref.n_eval[t]), int(r.n_eval[t]), "status", int(ref.status[t]), int(r.status[t]),
              "nll", ref.nll[t], r.nll[t], "dtheta", np.abs(ref.theta[t] - r.theta[t]).max(), flush=True)
