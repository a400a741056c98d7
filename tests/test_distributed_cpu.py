"""CPU, world_size 2, gloo: the N>1 path of bench.py -- shard tiles with no data-path collective, then ONE
gather of per-tile results back into the reference's tile order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpsat_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                 # same tile list on every rank
        T = 23
        N = rng.choice([64, 128, 500], size=T)
        P = rng.integers(0, 7, size=T)
        parts = sharding.partition_tiles(N, P, world)
        mine = parts[rank]
        # stand-in per-tile results that encode the global tile id (no compute in CPU tests)
        fixed = torch.tensor(np.stack([mine * 10.0 + k for k in range(4)], axis=1), dtype=torch.float32)
        cnt = P[mine]
        preds = torch.tensor(np.concatenate([np.full((c, 3), float(t)) + np.arange(c)[:, None] * 0.01
                                             for t, c in zip(mine, cnt)] + [np.zeros((0, 3))]), dtype=torch.float32)
        out = sharding.gather_results(fixed, preds, cnt, mine, world, rank)
        # equal-size variant used by bench.py
        fa, pa = sharding.all_gather_equal(torch.full((3, 2), float(rank)), torch.full((5, 3), float(rank) + 0.5), world)
        assert fa.shape == (3 * world, 2) and pa.shape == (5 * world, 3)
        assert fa[:3].eq(0).all() and fa[3:].eq(1).all() and pa[5:].eq(1.5).all()
        if rank == 0:
            fg, pg, off = out
            ok = True
            ok &= np.allclose(fg[:, 0], np.arange(T) * 10.0) and np.allclose(fg[:, 3], np.arange(T) * 10.0 + 3)
            ok &= off.tolist() == np.concatenate([[0], np.cumsum(P)]).tolist()
            for t in range(T):
                seg = pg[off[t]:off[t + 1]]
                ok &= seg.shape[0] == P[t] and np.allclose(seg[:, 0], t + np.arange(P[t]) * 0.01, atol=1e-5)
            ret[0] = bool(ok)
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] is True
