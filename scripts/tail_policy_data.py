"""Developer: per-tile evaluation counts of the bench workload at several iteration budgets (input of scripts/tail_policy_sim.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn

T, N, P, D = 4096, 500, 500, 3
b = syn.make_batch(T, N, P, D, 0, base_seed=0)
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
out = {}
for mi in (2, 4, 6, 20):
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=np.ones((T, D + 2)),
                              lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", max_iter=mi)
    out[f"n_eval_{mi}"] = r.n_eval.copy(); out[f"n_iter_{mi}"] = r.n_iter.copy(); out[f"status_{mi}"] = r.status.copy()
    print(mi, "evals/tile", r.n_eval.mean(), "max", r.n_eval.max(), "kernel_ms", r.kernel_ms, flush=True)
np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "tail_policy_data.npz"), **out)
