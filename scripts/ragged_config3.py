"""Developer: BASELINE configs[2] shape on one GPU -- mixed tile sizes N in {128..2048} (ragged CSR batch), Matern-3/2,
fp32, L-BFGS max_iter 20, P = 500.  Tiles are generated once per size class and replicated (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
T = int(os.environ.get("T", 1024))
P, D, kid = 500, 3, 2
sizes = [128, 256, 384, 512, 768, 1024, 1536, 2048]
rng = np.random.default_rng(0)
Ns = rng.choice(sizes, T)
proto = {n: [syn.make_tile(100 + 10 * i + j, n, P, D, kid) for j in range(2)] for i, n in enumerate(sizes)}
Xl, yl, Xsl = [], [], []
for t, n in enumerate(Ns):
    x_, y_, xs_, _ = proto[int(n)][t % 2]
    Xl.append(x_); yl.append(y_); Xsl.append(xs_)
X, y, Xs = np.concatenate(Xl).astype(np.float32), np.concatenate(yl).astype(np.float32), np.concatenate(Xsl).astype(np.float32)
obs_off = np.concatenate([[0], np.cumsum(Ns)]); pred_off = np.arange(T + 1) * P
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
kw = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=pred_off, Xs=dXs, theta0=np.ones((T, D + 2)), lo=lo, hi=hi,
          kernel="Matern32", optimiser="lbfgs", max_iter=20)
eng.fit_predict_batch(**kw)
r = eng.fit_predict_batch(**kw)
ne = r.n_eval.astype(np.float64)
Nf = Ns.astype(np.float64)
F = (ne * (Nf ** 3 + (3.5 * D + 9) * Nf ** 2) + Nf ** 2 * P).sum()
print(f"ragged Matern32 f32: {T} tiles (mean N {Ns.mean():.0f}, max {Ns.max()}) in {r.kernel_ms:.1f} ms -> "
      f"{T / r.kernel_ms * 1e3:.1f} tiles/s, mean evals {ne.mean():.1f}, {F / r.kernel_ms / 1e9:.2f} TFLOP/s "
      f"({F / r.kernel_ms / 1e9 / 157.3 * 100:.1f}% of fp32 MFMA peak); status counts {np.bincount(r.status)}")
for n in sizes:
    m = Ns == n
    print(f"  N={n}: {m.sum()} tiles, mean evals {ne[m].mean():.1f}")
