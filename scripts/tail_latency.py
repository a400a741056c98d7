"""Developer: latency of ONE objective + gradient evaluation of an N = 500 tile for the ways the end of a launch can run it:
4-wave build with a CU-mate (T = 512), alone on its CU (T = 256 on 256 workgroups), 8-wave build alone (T = 256), 8-wave build
with helpers (T = 128, 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GPSAT_DEVELOPER"] = "1"
import numpy as np
from gpsat_amd import synthetic as syn
from gpsat_amd.engine import Engine
N, D, P, kid = 500, 3, 0, 0
protos = [syn.make_tile(5 + j, N, P, D, kid) for j in range(8)]
def run(T, wg, label, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    X = np.concatenate([protos[t % 8][0] for t in range(T)]).astype(np.float32)
    y = np.concatenate([protos[t % 8][1] for t in range(T)]).astype(np.float32)
    th = np.stack([protos[t % 8][3] for t in range(T)])
    eng = Engine(0, workgroups_per_cu=wg)
    kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, X=X, y=y, pred_off=np.zeros(T + 1, np.int64), Xs=np.zeros((0, D), np.float32),
              theta0=th, kernel="RBF", optimiser="none", want_grad=True)
    eng.fit_predict_batch(**kw)
    ms = [eng.fit_predict_batch(**kw).kernel_ms for _ in range(5)]
    print(f"{label:60s} T {T:4d}: kernel ms {min(ms):.3f} (min of 5) -> {T / min(ms):.0f} evaluations/ms", flush=True)
    eng.close()
    for k in (env or {}):
        del os.environ[k]
run(512, 0, "4-wave build, two workgroups per CU")
run(256, 0, "4-wave build, 256 workgroups (GPSAT_DEBUG_GRID=256)", {"GPSAT_DEBUG_GRID": "256"}) if False else None
run(256, 1, "8-wave build, one tile per CU, no helper free")
run(128, 1, "8-wave build, 128 tiles: one helper each")
run(64, 1, "8-wave build, 64 tiles")
run(128, 1, "8-wave build, 128 tiles, cooperative tiles off", {"GPSAT_DEBUG_COOP": "0"})
