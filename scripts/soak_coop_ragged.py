#!/usr/bin/env python3
"""Developer: a ragged batch (BASELINE configs[2]'s mix, 8-wave build, helpers in the tail) run REPS times with cooperative
tiles on, every run compared with ONE run with them off: differing tiles are counted and described."""
import argparse
import os
os.environ.setdefault("GPSAT_DEVELOPER", "1")     # GPSAT_DEBUG_* knobs are read in developer mode only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                  # noqa: E402

T = int(os.environ.get("T", 1024))
REPS = int(os.environ.get("REPS", 4))
ns = argparse.Namespace(tiles=T, nobs=500, npred=100, dim=3, kernel="Matern32", optimiser="lbfgs", max_iter=20,
                        exact_iters=False, global_tiles=0, workload="configs2")
w = bench.build_workload(ns, 0, 1, 16)
from gpsat_amd.engine import Engine           # noqa: E402
eng = Engine(0)
kw = dict(D=3, obs_off=w["obs_off"], X=w["X"], y=w["y"], pred_off=w["pred_off"], Xs=w["Xs"], theta0=w["theta0"], lo=w["lo"], hi=w["hi"],
          kernel="Matern32", optimiser="lbfgs", max_iter=20)
os.environ["GPSAT_DEBUG_COOP"] = "0"
ref = eng.fit_predict_batch(**kw)
print("off: kernel_ms", round(ref.kernel_ms, 1), "evals", float(ref.n_eval.mean()), flush=True)
os.environ["GPSAT_DEBUG_COOP"] = "1"
os.environ["GPSAT_DEBUG_COOP_STATS"] = "1"
for rep in range(REPS):
    r = eng.fit_predict_batch(**kw)
    bad = np.nonzero((r.theta != ref.theta).any(axis=1) | (r.nll != ref.nll) | (r.n_eval != ref.n_eval) | (r.status != ref.status))[0]
    pbad = int((np.asarray(r.f_mean) != np.asarray(ref.f_mean)).sum() + (np.asarray(r.f_var) != np.asarray(ref.f_var)).sum())
    print("on: run", rep, "kernel_ms", round(r.kernel_ms, 1), "differing tiles", len(bad), "differing predictions", pbad, flush=True)
    for t in bad[:4]:
        print("    tile", int(t), "N", int(w["Ns"][t]), "n_eval", int(ref.n_eval[t]), int(r.n_eval[t]), "status", int(ref.status[t]), int(r.status[t]),
              "nll", ref.nll[t], r.nll[t], flush=True)
