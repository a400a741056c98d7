"""Developer: how evenly do the resident workgroups run out of work at the end of a bench launch?  (diagnostic build:
make -C gpsat_amd/csrc prof; GPSAT_DEBUG_SEG=0 shows the run-to-completion queue.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPSAT_LIB", os.path.join(ROOT, "gpsat_amd", "csrc", "libgpsat_hip_prof.so"))
import ctypes as C
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn, _lib

T, N, P, D = int(os.environ.get("T", 4096)), 500, 500, 3
b = syn.make_batch(64, N, P, D, 0, base_seed=5)
rep = T // 64
X, y, Xs = np.tile(b["X"], (rep, 1)).astype(np.float32), np.tile(b["y"], rep).astype(np.float32), np.tile(b["Xs"], (rep, 1)).astype(np.float32)
y = (y.reshape(T, N) * (1.0 + 0.02 * np.arange(T)[:, None] / T)).reshape(-1).astype(np.float32)      # distinct tiles
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
lo, hi = syn.default_bounds(T, D)
eng = Engine(0)
lib = _lib.get_lib()
kw = dict(D=D, obs_off=np.arange(T + 1) * N, X=dX, y=dy, pred_off=np.arange(T + 1) * P, Xs=dXs, theta0=np.ones((T, D + 2)),
          lo=lo, hi=hi, kernel="RBF", optimiser="lbfgs", max_iter=20)
eng.fit_predict_batch(**kw)
r = eng.fit_predict_batch(**kw)
buf = (C.c_ulonglong * 2048)()
lib.gpsat_debug_spans.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.gpsat_debug_spans(eng._h, buf)
v = np.array(buf[:], dtype=np.float64)
st, en = v[:1024], v[1024:]
m = en > 0
t0 = st[m].min()
end = (en[m] - t0) / 100.0                 # microseconds
print(f"{m.sum()} workgroups; kernel {r.kernel_ms:.2f} ms, {r.n_eval.mean():.2f} evaluations per tile (max {r.n_eval.max()})")
print(f"workgroups ran out of work at {end.min() / 1e3:.2f} .. {end.max() / 1e3:.2f} ms (mean {end.mean() / 1e3:.2f}); "
      f"idle at the end: {100 * (1 - end.mean() / end.max()):.1f} % of the launch")
print("percentiles of the exit time / last exit:", np.round(np.percentile(end, [1, 10, 25, 50, 75, 90]) / end.max(), 3))
pb = (C.c_ulonglong * 64)()
lib.gpsat_debug_profile(eng._h, pb)
pv = np.array(pb[:], dtype=np.float64).reshape(4, 16)
if pv[0, 14] > 0:
    print(f"time-sliced queue: the workgroups waited for a tile {100 * pv[0, 13] / (m.sum() * pv[0, 14]):.1f} % of the launch "
          f"(sum of the waits / workgroups x launch, s_memtime ticks)")
