"""Developer: one launch of a chosen phase mix on the bench tile shape, for rocprofv3 --pmc passes.
usage: pmc_phases.py MODE [T]   MODE: potrf | grad | predict | fit"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn

mode = sys.argv[1]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
N, D, P = int(os.environ.get("N", 500)), 3, 500
b = syn.make_batch(32, N, P, D, 0, base_seed=1)
rep = T // 32
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(np.tile(b[k], (rep, 1) if b[k].ndim == 2 else rep)).to(dev) for k in ("X", "y", "Xs"))
obs_off = np.arange(T + 1) * N
eng = Engine(0)
th0 = np.tile(b["truth"], (rep, 1))
kw = dict(D=D, obs_off=obs_off, X=dX, y=dy, theta0=th0, kernel="RBF")
nop, nox = np.zeros(T + 1, dtype=np.int64), dXs[:0].contiguous()
if mode == "potrf":
    r = eng.fit_predict_batch(pred_off=nop, Xs=nox, optimiser="none", **kw)
elif mode == "grad":
    r = eng.fit_predict_batch(pred_off=nop, Xs=nox, optimiser="none", want_grad=True, **kw)
elif mode == "predict":
    r = eng.fit_predict_batch(pred_off=np.arange(T + 1) * P, Xs=dXs, optimiser="none", **kw)
else:
    lo, hi = syn.default_bounds(T, D)
    r = eng.fit_predict_batch(pred_off=np.arange(T + 1) * P, Xs=dXs, optimiser="lbfgs", max_iter=20, lo=lo, hi=hi,
                              **dict(kw, theta0=np.ones((T, D + 2))))
print(mode, "T", T, "kernel_ms", round(r.kernel_ms, 3), "evals", float(r.n_eval.mean()))
