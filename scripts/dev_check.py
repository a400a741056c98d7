"""Developer smoke: HIP path vs oracle on a few small tiles (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

eng = Engine(0)
print("device:", eng.device_name, flush=True)
names = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}
for (N, P, D, kid) in [(20, 5, 1, 0), (32, 8, 3, 0), (50, 40, 3, 2), (100, 33, 2, 3), (200, 64, 3, 1), (500, 100, 3, 0)]:
    b = syn.make_batch(3, N, P, D, kid, base_seed=7)
    T = 3
    th0 = np.tile(np.concatenate([np.full(D, 2.5), [0.5, 0.1]]), (T, 1))
    t0 = time.time()
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                              theta0=th0, kernel=names[kid], optimiser="none", want_grad=True)
    dt = time.time() - t0
    for t in range(T):
        a, e = b["obs_off"][t], b["obs_off"][t + 1]
        pa, pe = b["pred_off"][t], b["pred_off"][t + 1]
        Xd, yd, Xsd = b["X"][a:e].astype(np.float64), b["y"][a:e].astype(np.float64), b["Xs"][pa:pe].astype(np.float64)
        nll, g = go.nll_and_grad(kid, Xd, yd, th0[t])
        f, fv, yv = go.predict(kid, Xd, yd, Xsd, th0[t])
        print(f"N={N} P={P} D={D} k={kid} t={t} status={r.status[t]} nll {r.nll[t]:.6f} vs {nll:.6f} "
              f"| grad relerr {np.max(np.abs(r.grad[t]-g)/(np.abs(g)+1e-3)):.2e} "
              f"| f* err {np.max(np.abs(r.f_mean[pa:pe]-f)):.2e} var err {np.max(np.abs(r.f_var[pa:pe]-fv)):.2e} "
              f"| {dt*1e3:.1f} ms kernel {r.kernel_ms:.2f} ms", flush=True)
# optimiser
for (N, P, D, kid) in [(100, 16, 3, 0), (500, 64, 3, 0)]:
    T = 4
    b = syn.make_batch(T, N, P, D, kid, base_seed=11)
    th0 = np.ones((T, D + 2))
    lo, hi = syn.default_bounds(T, D)
    for opt, mi in (("lbfgs", 200), ("adam", 20)):
        r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                                  theta0=th0, lo=lo, hi=hi, kernel=names[kid], optimiser=opt, max_iter=mi)
        print(opt, "N", N, "status", r.status, "n_eval", r.n_eval, "kernel ms", round(r.kernel_ms, 2))
        print("  theta", np.round(r.theta[0], 4), "nll", r.nll)
    o = go.fit_predict_batch(kid, D, b["obs_off"], b["X"].astype(np.float64), b["y"].astype(np.float64), b["pred_off"],
                             b["Xs"].astype(np.float64), th0, lo, hi, np.ones(D + 2, bool), max_iter=200)
    print("oracle theta", np.round(o["theta"][0], 4), "nll", o["nll"], "n_eval", o["n_eval"])
    print("  max |f* diff|", np.max(np.abs(o["f_mean"] - r.f_mean)))
