#!/usr/bin/env python3
"""One rank of the RCCL rehearsal (tests/test_gpu_0_rccl.py starts it through torch.distributed.run; also usable by hand:
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/rccl_rank.py out.json`).

With a "nccl" (= RCCL) process group: the orchestrator's tile-sharded run through sharding.gather_arrays, compared on
rank 0 with the un-grouped run of the same configuration in the same process; then a raw sharding.gather_results of
device tensors.  Writes a JSON verdict on rank 0."""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid_problem(n_side=12, m=9000, seed=3):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 1, (m, 2))
    t = rng.uniform(-4, 4, m)
    z = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.05 * t + 0.1 * rng.standard_normal(m)
    df = pd.DataFrame({"x": xy[:, 0], "y": xy[:, 1], "t": t, "z": z})
    g = (np.arange(n_side) + 0.5) / n_side
    xl = pd.DataFrame([(a, b, 0.0) for a in g for b in g], columns=["x", "y", "t"])
    return dict(
        expert_loc_config={"source": xl},
        data_config={"data_source": df, "obs_col": "z", "coords_col": ["x", "y", "t"],
                     "local_select": [{"col": ["x", "y"], "comp": "<", "val": 0.08},
                                      {"col": "t", "comp": "<=", "val": 4}, {"col": "t", "comp": ">=", "val": -4}]},
        model_config={"oi_model": "HipGPRModel",
                      "init_params": {"kernel": "Matern32", "obs_mean": "local", "coords_scale": [0.05, 0.05, 1.0]},
                      "constraints": {"lengthscales": {"low": [1e-8, 1e-8, 1e-8], "high": [12.0, 12.0, 9.0]}},
                      "optim_kwargs": {"max_iter": 20}},
        pred_loc_config={"method": "shift_arrays", "x": np.array([-0.01, 0.0, 0.01]), "y": np.array([0.0, 0.01])})


def main():
    out_path = sys.argv[1]
    import torch
    import torch.distributed as dist
    from gpsat_amd import sharding
    from gpsat_amd.engine import Engine
    from gpsat_amd.local_experts import BatchedLocalExpertOI

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    verdict = {"backend": dist.get_backend(), "world": world, "device": torch.cuda.get_device_name(local)}
    eng = Engine(local)
    cfg = grid_problem()
    grouped = BatchedLocalExpertOI(engine=eng, **cfg).run(gather="always")             # rank / world from the group
    dist.barrier()
    if rank == 0:
        plain = BatchedLocalExpertOI(engine=eng, **cfg).run(rank=0, world_size=1, gather=False)
        same = set(grouped) == set(plain)
        for k in plain:
            a = grouped[k].drop(columns=[c for c in ("run_time", "config_id") if c in grouped[k].columns])
            b = plain[k].drop(columns=[c for c in ("run_time", "config_id") if c in plain[k].columns])
            same = same and a.equals(b)
        verdict.update(tables_equal=bool(same), experts=int(len(plain["run_details"])), preds=int(len(plain["preds"])))
    # raw gather of device tensors (what bench.py --global-tiles does per step)
    T = 37
    parts = sharding.partition_tiles(np.full(T, 100), np.arange(T) % 5, world)
    mine = parts[rank]
    cnt = (np.arange(T) % 5)[mine]
    fixed = torch.tensor(np.stack([mine * 10.0 + k for k in range(4)], axis=1), dtype=torch.float64, device=dev)
    preds = torch.tensor(np.concatenate([np.full((c, 3), float(t)) for t, c in zip(mine, cnt)] + [np.zeros((0, 3))]),
                         dtype=torch.float32, device=dev)
    got = sharding.gather_results(fixed, preds, cnt, mine, world, rank, device=dev, total=T)
    tm = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    if rank == 0:
        fg, pg, off = got
        ok = np.allclose(fg[:, 0], np.arange(T) * 10.0) and off[-1] == int((np.arange(T) % 5).sum())
        ok = ok and all(np.all(pg[off[t]:off[t + 1], 0] == t) for t in range(T))
        verdict.update(raw_gather_ok=bool(ok), all_reduce_max=float(tm.item()))
        with open(out_path, "w") as f:
            json.dump(verdict, f)
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
