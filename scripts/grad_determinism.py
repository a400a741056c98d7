#!/usr/bin/env python3
"""Developer: objective + gradient of configs[1]'s 4096 tiles at fixed parameters, REPS times -- are the runs of one
library build (GPSAT_LIB) bit-identical, and how far is each from the first run of the shipped build (saved to / read from
gpurun_out/grad_ref.npz)?"""
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
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with threadpool_limits(1):
    with ThreadPoolExecutor(16) as pool:
        tiles = list(pool.map(lambda t: syn.make_tile(t, N, P, D, kid), range(T)))
X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
rng = np.random.default_rng(5)
th = np.exp(rng.normal(0.0, 0.5, (T, D + 2)))
kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
          theta0=th, kernel="RBF", optimiser="none", want_grad=True)
eng = Engine(0)
tag = os.path.basename(os.environ.get("GPSAT_LIB", "default"))
ref_path = os.path.join(ROOT, "gpurun_out", "grad_ref.npz")
first = None
for rep in range(REPS):
    r = eng.fit_predict_batch(X=X, y=y, Xs=Xs, **kw)
    g = r.grad.copy()
    if first is None:
        first = g
        first_nll, first_f = r.nll.copy(), r.f_mean.copy()
        if os.path.exists(ref_path):
            ref = np.load(ref_path)["g"]
            rel = np.abs(g - ref) / (np.abs(ref) + 1e-3 * np.abs(ref).max(axis=1, keepdims=True))
            print(tag, "vs saved reference: max rel", float(rel.max()), "tiles > 1e-3:", np.nonzero(rel.max(axis=1) > 1e-3)[0][:10].tolist())
        else:
            np.savez(ref_path, g=g)
            print(tag, "reference saved")
    else:
        bad = np.nonzero((g != first).any(axis=1))[0]
        rel = np.abs(g - first) / (np.abs(first) + 1e-30)
        print(tag, "rep", rep, "tiles differing from rep 0:", len(bad), bad[:10].tolist(), "max rel diff", float(rel.max()),
              "| nll differs in", int((r.nll != first_nll).sum()), "tiles, predictions in", int((r.f_mean != first_f).reshape(T, P).any(axis=1).sum()),
              "| components differing:", (g != first).sum(axis=0).tolist(), flush=True)
eng.close()
