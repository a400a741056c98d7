#!/usr/bin/env python3
"""Developer: objective / gradient error of the fp32 kernels against the fp64 oracle at fixed parameters, large tiles
(GPSAT_LIB selects the library build)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402
from oracle import gp_oracle as go       # noqa: E402

eng = Engine(0)
for kid, N in ((2, 512), (2, 1024), (2, 2048), (0, 500), (0, 2048)):
    D, P = 3, 8
    X, y, Xs, truth = syn.make_tile(1234 + N, N, P, D, kid)
    for th in (truth * np.array([1.3, 0.8, 1.1, 2.0, 1.5]), np.array([1.0, 1.0, 1.0, 1.0, 1.0]), truth):
        r = eng.fit_predict_batch(D=D, obs_off=[0, N], X=X.astype(np.float32), y=y.astype(np.float32), pred_off=[0, P],
                                  Xs=Xs.astype(np.float32), theta0=th[None, :], kernel=kid, optimiser="none", want_grad=True)
        f, g = go.nll_and_grad(kid, X.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64), th)
        print(os.path.basename(os.environ.get("GPSAT_LIB", "default")), "kid", kid, "N", N, "dnll", f"{r.nll[0] - f:+.3e}",
              "grad rel err", np.array2string(np.abs(r.grad[0] - g) / (np.abs(g) + 1e-12), precision=2), "g", np.array2string(g, precision=3))
