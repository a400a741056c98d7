"""Developer: randomised fit parity on the GPU box -- converged L-BFGS fits (fp64 kernels, tight tolerances) against the
oracle's SciPy L-BFGS-B fits from the same start; counts tiles whose optimum objective differs (different basin)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

names = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}
rng = np.random.default_rng(int(os.environ.get("SEED", 3)))
eng = Engine(0)
n_cases = int(os.environ.get("CASES", 12))
tot = worse = better = same = 0
t0 = time.time()
for case in range(n_cases):
    D = int(rng.integers(1, 4)); kid = int(rng.integers(0, 4)); T = 8
    Ns = [int(rng.integers(40, 260)) for _ in range(T)]
    b = syn.make_batch(T, Ns, 4, D, kid, base_seed=int(rng.integers(0, 10**6)), dtype=np.float64)
    lo, hi = syn.default_bounds(T, D)
    th0 = np.ones((T, D + 2))
    for dtype, ftol, tol in (("f64", 1e-12, 1e-5), ("f32", 1e-9, 3e-4)):
        r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"].astype(np.float32 if dtype == "f32" else np.float64),
                                  y=b["y"].astype(np.float32 if dtype == "f32" else np.float64), pred_off=b["pred_off"],
                                  Xs=b["Xs"].astype(np.float32 if dtype == "f32" else np.float64), theta0=th0, lo=lo, hi=hi,
                                  kernel=names[kid], optimiser="lbfgs", max_iter=500, ftol=ftol, dtype=dtype)
        o = go.fit_predict_batch(kid, D, b["obs_off"], b["X"], b["y"], b["pred_off"], b["Xs"], th0, lo, hi,
                                 np.ones(D + 2, bool), max_iter=500)
        for t in range(T):
            # objective of the GPU optimum re-evaluated by the oracle (fp64) against the oracle's own optimum
            a, e = b["obs_off"][t], b["obs_off"][t + 1]
            f_gpu, _ = go.nll_and_grad(kid, b["X"][a:e], b["y"][a:e], r.theta[t])
            d = f_gpu - o["nll"][t]
            tot += 1
            if abs(d) <= tol * Ns[t]:
                same += 1
            elif d < 0:
                better += 1
            else:
                worse += 1
                print(f"  case {case} tile {t} {dtype} D={D} {names[kid]} N={Ns[t]}: GPU optimum higher by {d:.3e} "
                      f"(status {r.status[t]}, evals {r.n_eval[t]} vs {o['n_eval'][t]})", flush=True)
    print(f"case {case + 1}/{n_cases} {time.time() - t0:.0f}s: same {same} better {better} worse {worse} of {tot}", flush=True)
