"""Developer: per-phase kernel time split on the GPU box (objective only / objective+gradient / predict)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn

T, N, D = int(os.environ.get("T", 2048)), int(os.environ.get("N", 500)), 3
b = syn.make_batch(32, N, 500, D, 0, base_seed=1)
rep = T // 32
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
obs_off = np.arange(T + 1) * N
th0 = np.tile(b["truth"], (rep, 1))
lo, hi = syn.default_bounds(T, D)
for wg in [int(v) for v in os.environ.get("WGS", "1,2,3,4").split(",")]:
    eng = Engine(0, workgroups_per_cu=wg)
    res = {}
    for name, P, kw in [("potrf", 0, dict(optimiser="none")),
                        ("potrf+trtri+grad", 0, dict(optimiser="none", want_grad=True)),
                        ("potrf+predict500", 500, dict(optimiser="none")),
                        ("lbfgs20+predict500", 500, dict(optimiser="lbfgs", max_iter=20, theta0=np.ones((T, D + 2)), lo=lo, hi=hi))]:
        pred_off = np.arange(T + 1) * P
        args = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=pred_off, Xs=dXs[:T * P].contiguous() if P else dXs[:0].contiguous(),
                    theta0=th0, kernel="RBF")
        args.update(kw)
        eng.fit_predict_batch(**args)
        r = eng.fit_predict_batch(**args)
        res[name] = r.kernel_ms
        extra = f" evals/tile {r.n_eval.mean():.1f} conv {np.mean(r.status==0):.2f}" if "lbfgs" in name else ""
        print(f"wg/cu={wg} {name:22s} {r.kernel_ms:9.2f} ms  -> {r.kernel_ms*1e3/T:8.2f} us/tile{extra}", flush=True)
    eng.close()
