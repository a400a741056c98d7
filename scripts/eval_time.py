"""Developer: kernel time of ONE objective + gradient evaluation per tile at fixed parameters (optimiser none), for A/B of builds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd import synthetic as syn
from gpsat_amd.engine import Engine
for T, N, wg in ((4096, 500, 0), (1024, 1024, 0), (512, 2048, 0)):
    D, P, kid = 3, 0, 2
    protos = [syn.make_tile(5 + j, N, P, D, kid) for j in range(8)]
    X = np.concatenate([protos[t % 8][0] for t in range(T)]).astype(np.float32)
    y = np.concatenate([protos[t % 8][1] for t in range(T)]).astype(np.float32)
    th = np.stack([protos[t % 8][3] for t in range(T)])
    eng = Engine(0, workgroups_per_cu=wg)
    kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, X=X, y=y, pred_off=np.zeros(T + 1, np.int64), Xs=np.zeros((0, D), np.float32),
              theta0=th, kernel="Matern32", optimiser="none", want_grad=True)
    eng.fit_predict_batch(**kw)
    ms = [eng.fit_predict_batch(**kw).kernel_ms for _ in range(3)]
    print(os.path.basename(os.environ.get("GPSAT_LIB", "base")), f"T {T} N {N}: kernel ms", [round(m, 2) for m in ms], flush=True)
    eng.close()
