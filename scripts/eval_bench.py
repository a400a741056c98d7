"""Developer: objective + gradient evaluations per second (one evaluation per tile, no optimiser), for kernel experiments
whose results need not be right (e.g. operand loads redirected): GPSAT_LIB selects the library variant."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn

T, N, D = int(os.environ.get("T", 16384)), int(os.environ.get("N", 500)), 3
b = syn.make_batch(32, N, 500, D, 0, base_seed=1)
rep = T // 32
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
eng = Engine(0, workgroups_per_cu=int(os.environ.get("WG", 2)))
args = dict(D=D, obs_off=np.arange(T + 1) * N, X=dX, y=dy, pred_off=np.zeros(T + 1, dtype=np.int64), Xs=dXs[:0].contiguous(),
            theta0=np.tile(b["truth"], (rep, 1)), kernel="RBF", optimiser="none", want_grad=os.environ.get("GRAD", "1") == "1")
eng.fit_predict_batch(**args)
ms = [eng.fit_predict_batch(**args).kernel_ms for _ in range(3)]
print(f"{os.path.basename(os.environ.get('GPSAT_LIB', 'default'))}: T={T} N={N} kernel {min(ms):.2f} ms -> {T / min(ms):.1f} k evaluations/s")
