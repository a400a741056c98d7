"""Developer: BASELINE configs[4] shape on one GPU -- fp64, N=2000 obs/tile, predict-only with given hyper-parameters."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
T, N, P, D = int(os.environ.get("T", 256)), 2000, 500, 3
b = syn.make_batch(8, N, P, D, 0, base_seed=1, dtype=np.float64)
rep = T // 8
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
eng = Engine(0)
kw = dict(D=D, obs_off=np.arange(T + 1) * N, X=dX, y=dy, pred_off=np.arange(T + 1) * P, Xs=dXs,
          theta0=np.tile(b["truth"], (rep, 1)), kernel="RBF", optimiser="none", dtype="f64")
eng.fit_predict_batch(**kw)
r = eng.fit_predict_batch(**kw)
F = N ** 3 / 3 + N * N * P
print(f"fp64 N={N} P={P} predict-only: {T} tiles in {r.kernel_ms:.1f} ms -> {T / r.kernel_ms * 1e3:.1f} tiles/s, "
      f"{T * F / r.kernel_ms / 1e9:.2f} TFLOP/s ({T * F / r.kernel_ms / 1e9 / 78.6 * 100:.1f}% of fp64 MFMA peak)")
