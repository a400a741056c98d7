"""Developer: what the prediction of a tile costs beside its factorisation: kernel time of 4096 N = 500 tiles at fixed parameters
(optimiser none: one factorisation, no gradient) with P = 0 / 100 / 500 prediction points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd import synthetic as syn
from gpsat_amd.engine import Engine
T, N, D, kid = 4096, 500, 3, 0
eng = Engine(0)
for P in (0, 100, 500):
    protos = [syn.make_tile(5 + j, N, max(P, 1), D, kid) for j in range(8)]
    X = np.concatenate([protos[t % 8][0] for t in range(T)]).astype(np.float32)
    y = np.concatenate([protos[t % 8][1] for t in range(T)]).astype(np.float32)
    Xs = np.concatenate([protos[t % 8][2][:P] for t in range(T)]).astype(np.float32).reshape(-1, D)
    th = np.stack([protos[t % 8][3] for t in range(T)])
    kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, X=X, y=y, pred_off=np.arange(T + 1, dtype=np.int64) * P, Xs=Xs,
              theta0=th, kernel="RBF", optimiser="none")
    eng.fit_predict_batch(**kw)
    ms = min(eng.fit_predict_batch(**kw).kernel_ms for _ in range(4))
    fl = T * (N ** 3 / 3 + N * N * P) * 1e-12 if P else T * (N ** 3 / 3) * 1e-12
    print(f"P = {P:3d}: kernel {ms:7.3f} ms  ({T / ms:.0f} tiles/ms)", flush=True)
eng.close()
