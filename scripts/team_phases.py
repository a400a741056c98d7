#!/usr/bin/env python3
"""Developer: where a team's time goes -- one evaluation of one fp64 tile with and without the gradient half, with and without
prediction points, team sizes 1 / 2 / 4 / 8 / 16 (kernel ms)."""
import os
os.environ.setdefault("GPSAT_DEVELOPER", "1")     # GPSAT_DEBUG_* knobs are read in developer mode only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402

eng = Engine(0)
for N in (2000, 2500):
    for P, grad in ((0, False), (0, True), (100, False)):
        b = syn.make_batch(1, [N], P, 3, 0, base_seed=3, dtype=np.float64)
        row = []
        for g in (1, 4, 8, 16, 32):
            os.environ["GPSAT_DEBUG_TEAM"] = str(g)
            ms = min(eng.fit_predict_batch(D=3, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                                           theta0=b["truth"], kernel="RBF", optimiser="none", want_grad=grad, dtype="f64").kernel_ms
                     for _ in range(3))
            row.append(f"G={g}: {ms:.2f}")
        print(f"N={N} P={P} grad={grad}: " + "  ".join(row), flush=True)
