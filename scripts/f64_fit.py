"""Developer: fp64 fit + predict throughput (the reference's native precision) on the BASELINE configs[1] shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
T, N, P, D = int(os.environ.get("T", 4096)), int(os.environ.get("N", 500)), 500, 3
b = syn.make_batch(32, N, P, D, 0, base_seed=1, dtype=np.float64)
rep = T // 32
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
kw = dict(D=D, obs_off=np.arange(T + 1) * N, X=dX, y=dy, pred_off=np.arange(T + 1) * P, Xs=dXs, theta0=np.ones((T, D + 2)),
          lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", max_iter=20, dtype="f64")
eng.fit_predict_batch(**kw)
r = eng.fit_predict_batch(**kw)
E = r.n_eval.mean()
F = T * (E * (N ** 3 + (3.5 * D + 9) * N * N) + N * N * P)
print(f"fp64 fit N={N} P={P} max_iter=20: {T} tiles in {r.kernel_ms:.1f} ms -> {T / r.kernel_ms * 1e3:.1f} tiles/s, "
      f"{E:.1f} evals/tile, {F / r.kernel_ms / 1e9:.2f} TFLOP/s ({F / r.kernel_ms / 1e9 / 78.6 * 100:.1f}% of fp64 MFMA peak), "
      f"status {np.bincount(r.status)}")
