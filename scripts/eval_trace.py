"""Developer: evaluations spent per L-BFGS iteration on the bench workload (no oracle): the same tiles are run with
max_iter = 1..20 and n_eval(max_iter) differenced."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
T, N, P, D = 128, 500, 0, 3
b = syn.make_batch(T, N, P, D, 0, base_seed=1_000_000)
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
ne = np.zeros((21, T)); st = np.zeros((21, T), int); nll = np.zeros((21, T))
for mi in range(1, 21):
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                              theta0=np.ones((T, D + 2)), lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", max_iter=mi)
    ne[mi], st[mi], nll[mi] = r.n_eval, r.status, r.nll
d = np.diff(ne, axis=0)      # evals spent in iteration mi (rows 0..19 -> iterations 1..20)
print("mean evals per iteration (over tiles still running):")
for mi in range(1, 21):
    running = st[mi - 1] != 0 if mi > 1 else np.ones(T, bool)
    print(f"  it {mi:2d}: {d[mi - 1][running].mean():.2f} evals, running {running.sum():3d}, converged so far {(st[mi] == 0).sum():3d}, "
          f"mean nll {nll[mi].mean():.3f}")
print("total evals/tile at max_iter=20:", ne[20].mean(), " histogram:", np.bincount(ne[20].astype(int))[10:45])
