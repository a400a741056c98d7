"""Developer: is the number of evaluations a tile needs predictable from cheap statistics (for longest-first ordering)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
T, N, P, D = 1024, 500, 0, 3
b = syn.make_batch(T, N, P, D, 0, base_seed=1_000_000)
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
kw = dict(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel="RBF")
r = eng.fit_predict_batch(optimiser="lbfgs", max_iter=20, **kw)
r0 = eng.fit_predict_batch(optimiser="none", want_grad=True, **kw)
y = b["y"].reshape(T, N)
feats = {"var(y)": y.var(1), "nll(theta0)": r0.nll, "|grad|": np.linalg.norm(r0.grad, axis=1), "grad_sf": r0.grad[:, D], "grad_sn": r0.grad[:, D + 1],
         "grad_l0": r0.grad[:, 0], "truth l min": b["truth"][:, :D].min(1), "truth l max": b["truth"][:, :D].max(1)}
ne = r.n_eval.astype(float)
print("n_eval mean", ne.mean(), "std", ne.std())
for k, v in feats.items():
    print(f"  corr(n_eval, {k}) = {np.corrcoef(ne, v)[0, 1]:+.3f}")
