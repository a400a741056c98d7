import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from gpsat_amd import synthetic as syn
from gpsat_amd.engine import Engine
eng = Engine(0)
os.environ["GPSAT_DEBUG_TEAM_STATS"] = "1"
for N in (2500,):
    b = syn.make_batch(1, [N], 0, 3, 0, base_seed=3, dtype=np.float64)
    for g in (2, 8):
        os.environ["GPSAT_DEBUG_TEAM"] = str(g)
        for _ in range(2):
            r = eng.fit_predict_batch(D=3, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=b["truth"], kernel="RBF", optimiser="none", dtype="f64")
        print("N", N, "G", g, "kernel_ms", r.kernel_ms, flush=True)
