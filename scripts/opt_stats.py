"""Developer: optimiser efficiency (evaluations per tile, objective reached) vs the scipy L-BFGS-B oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

T, N, P, D = 64, 500, 8, 3
b = syn.make_batch(T, N, P, D, 0, base_seed=1_000_000)
th0 = np.ones((T, D + 2)); lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
o20 = go.fit_predict_batch(0, D, b["obs_off"], b["X"].astype(np.float64), b["y"].astype(np.float64), b["pred_off"],
                           b["Xs"].astype(np.float64), th0, lo, hi, np.ones(D + 2, bool), max_iter=20)
oc = go.fit_predict_batch(0, D, b["obs_off"], b["X"].astype(np.float64), b["y"].astype(np.float64), b["pred_off"],
                          b["Xs"].astype(np.float64), th0, lo, hi, np.ones(D + 2, bool), max_iter=1000)
print("oracle maxiter=20: evals/tile %.1f  mean(nll - nll_conv) %.4f  max %.4f" % (o20["n_eval"].mean(), (o20["nll"] - oc["nll"]).mean(), (o20["nll"] - oc["nll"]).max()))
print("oracle converged: evals/tile %.1f" % oc["n_eval"].mean())
for kw in [dict(max_iter=20), dict(max_iter=20, max_ls=6), dict(max_iter=20, max_ls=3), dict(max_iter=20, ftol=1e-6), dict(max_iter=20, ftol=1e-5),
           dict(max_iter=1000), dict(max_iter=1000, ftol=1e-6)]:
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=th0,
                              lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", **kw)
    d = r.nll - oc["nll"]
    print(kw, "evals/tile %.1f (min %d max %d) conv %.2f  mean(nll-nll_conv) %.4f max %.4f" % (r.n_eval.mean(), r.n_eval.min(), r.n_eval.max(), np.mean(r.status == 0), d.mean(), d.max()))
    print("    n_eval hist:", np.bincount(np.minimum(r.n_eval, 60) // 5))
