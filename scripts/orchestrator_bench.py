"""Developer: end-to-end throughput of the batched orchestrator (selection + packing + fit/predict + tables + store flush)
on a synthetic sweep shaped like BASELINE configs[1]: T expert locations on a grid, ~500 observations per tile (x, y, t),
RBF, 20 optimiser steps, P prediction locations per tile from a grid."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from gpsat_amd.engine import Engine
from gpsat_amd.local_experts import BatchedLocalExpertOI

side = int(os.environ.get("SIDE", 64))                      # side*side expert locations
dens = float(os.environ.get("NOBS", 500))
rng = np.random.default_rng(0)
r_obs = 0.75 / side * 2                                     # selection radius in units of the domain
M = int(dens / (np.pi * r_obs ** 2))                        # points so that a ball holds ~NOBS of them
df = pd.DataFrame({"x": rng.uniform(0, 1, M), "y": rng.uniform(0, 1, M), "t": rng.uniform(18000, 18008, M)})
df["z"] = np.sin(9 * df["x"]) * np.cos(7 * df["y"]) + 0.05 * (df["t"] - 18004) + 0.1 * rng.standard_normal(M)
g = (np.arange(side) + 0.5) / side
xl = pd.DataFrame([(a, b, 18004.0) for a in g for b in g], columns=["x", "y", "t"])
pg = (np.arange(4 * side) + 0.5) / (4 * side)
pred = pd.DataFrame([(a, b) for a in pg for b in pg], columns=["x", "y"])
cfg = dict(
    expert_loc_config={"source": xl},
    data_config={"data_source": df, "obs_col": "z", "coords_col": ["x", "y", "t"],
                 "local_select": [{"col": ["x", "y"], "comp": "<", "val": r_obs}, {"col": "t", "comp": "<=", "val": 4},
                                  {"col": "t", "comp": ">=", "val": -4}]},
    model_config={"oi_model": "HipGPRModel", "init_params": {"kernel": "RBF", "obs_mean": "local",
                                                             "coords_scale": [r_obs / 6, r_obs / 6, 1.0]},
                  "constraints": {"lengthscales": {"low": [1e-8] * 3, "high": [12 * r_obs / 6, 12 * r_obs / 6, 9.0]}},
                  "optim_kwargs": {"max_iter": 20}},
    pred_loc_config={"method": "from_dataframe", "df": pred, "max_dist": 1.0 / side})
eng = Engine(0)
chunk = int(os.environ.get("CHUNK", 0)) or None             # tiles per engine call (None: the orchestrator's default)
se = int(os.environ.get("STORE_EVERY", 4096))                # expert locations per flushed wave
for dev_sel in ((True,) if os.environ.get("DEVICE_ONLY") else (True, False)):
    with tempfile.TemporaryDirectory() as d:
        oi = BatchedLocalExpertOI(engine=eng, device_select=dev_sel, **cfg)
        if os.environ.get("ENGINES"):
            oi.engine_workers = int(os.environ["ENGINES"])
        if dev_sel:
            oi.run(store_path=os.path.join(d, "warm"), store_every=se, engine_chunk=chunk)      # warm-up (allocations, first launch)
        t0 = time.perf_counter()
        tabs = oi.run(store_path=os.path.join(d, "s"), store_every=se, engine_chunk=chunk)
        dt = time.perf_counter() - t0
        rd = tabs["run_details"]
        print(f"device_select={dev_sel}: {len(rd)} experts, mean obs/tile {rd['num_obs'].mean():.0f}, preds {len(tabs['preds'])}, "
              f"{dt:.3f} s end to end -> {len(rd) / dt:.0f} tiles/s; split (s): " + ", ".join(f"{k} {v:.3f}" for k, v in oi.timings.items() if k != "calls")
              + f"; calls (job, tiles, start, end, kernel s): {oi.timings['calls']}")
