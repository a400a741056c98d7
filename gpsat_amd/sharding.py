"""Tile sharding across GPUs (one process per GPU) and the final gather.

Tiles are independent: the tile list is split by a longest-processing-time
greedy on the cost model c_t = E*N_t^3 + N_t^2*P_t, every rank runs its own packed batch with NO
data-path collective, and one gather at the end returns per-tile hyper-parameters and predictions
to rank 0 in the reference's tile order.  Works on any torch.distributed backend ("nccl" = RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def tile_cost(N, P, n_eval=20):
    N = np.asarray(N, dtype=np.float64)
    P = np.asarray(P, dtype=np.float64)
    return n_eval * N ** 3 + N * N * P


def partition_tiles(N, P, world_size, n_eval=20):
    """LPT greedy.  Returns a list (len world_size) of int64 index arrays, each sorted ascending so a
    rank keeps the reference's relative tile order.  Deterministic."""
    cost = tile_cost(N, P, n_eval)
    order = np.argsort(-cost, kind="stable")
    loads = np.zeros(world_size)
    bins = [[] for _ in range(world_size)]
    for t in order:
        r = int(np.argmin(loads))          # first minimum -> deterministic
        bins[r].append(int(t))
        loads[r] += cost[t]
    return [np.array(sorted(b), dtype=np.int64) for b in bins]


def pack_subset(batch, idx):
    """Sub-batch (CSR re-packed) of a dict produced by synthetic.make_batch-style packing."""
    obs_off, pred_off = batch["obs_off"], batch["pred_off"]
    D = batch["D"]
    Ns = (obs_off[1:] - obs_off[:-1])[idx]
    Ps = (pred_off[1:] - pred_off[:-1])[idx]
    rows = np.concatenate([np.arange(obs_off[t], obs_off[t + 1]) for t in idx]) if len(idx) else np.zeros(0, np.int64)
    prow = np.concatenate([np.arange(pred_off[t], pred_off[t + 1]) for t in idx]) if len(idx) else np.zeros(0, np.int64)
    out = dict(batch)
    out.update(T=len(idx), obs_off=np.concatenate([[0], np.cumsum(Ns)]).astype(np.int64),
               pred_off=np.concatenate([[0], np.cumsum(Ps)]).astype(np.int64),
               X=batch["X"][rows].reshape(-1, D), y=batch["y"][rows], Xs=batch["Xs"][prow].reshape(-1, D))
    for k in ("truth",):
        if k in batch:
            out[k] = batch[k][idx]
    return out


def assemble_global(shards, total=None):
    """Merge per-shard results into the reference's (global) tile order.

    shards: iterable of (fixed [T_r, F], preds [sumP_r, C], pred_counts [T_r], tile_index [T_r]) numpy arrays, one per
    shard; tile_index holds the GLOBAL ids of the shard's tiles in the order its rows are packed.
    Returns (fixed_global [total, F] with NaN rows for tiles no shard owns, preds_global [sumP, C] in global tile order,
    pred_off_global [total+1]).  Pure host code: the same routine closes the RCCL gather (``gather_results``) and a
    single-process run over logical shards."""
    shards = [tuple(np.asarray(a) for a in sh) for sh in shards]
    if total is None:
        total = int(sum(len(sh[3]) for sh in shards))
    F = shards[0][0].shape[1]
    C = shards[0][1].shape[1] if shards[0][1].ndim == 2 else 3
    fixed_g = np.full((total, F), np.nan)
    counts_g = np.zeros(total, dtype=np.int64)
    for fx, _, cnt, ids in shards:
        ids = ids.astype(np.int64)
        assert len(fx) == len(ids) == len(cnt)
        fixed_g[ids] = fx
        counts_g[ids] = cnt
    pred_off_g = np.concatenate([[0], np.cumsum(counts_g)]).astype(np.int64)
    pdt = np.result_type(*[sh[1].dtype for sh in shards])
    preds_g = np.zeros((int(pred_off_g[-1]), C), dtype=pdt)
    for _, pr, cnt, ids in shards:
        cnt = cnt.astype(np.int64)
        if cnt.sum() == 0:
            continue
        src_off = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        # destination row of every prediction row of this shard
        dst = np.repeat(pred_off_g[ids.astype(np.int64)] - src_off, cnt) + np.arange(int(cnt.sum()))
        preds_g[dst] = pr[:int(cnt.sum())]
    return fixed_g, preds_g, pred_off_g


def gather_results(fixed, preds, pred_counts, tile_index, world_size, rank, device=None, dst=0, total=None):
    """ONE gather(v) of per-tile results to ``dst``: nothing is sent to the other ranks.

    fixed       : [T_r, F] torch tensor (theta, nll, status, n_eval ... per local tile)
    preds       : [sumP_r, C] torch tensor (f*, f*_var, y_var)
    pred_counts : [T_r] int64 numpy, predictions per local tile
    tile_index  : [T_r] int64 numpy, global tile ids of the local tiles
    Returns on dst: (fixed_global [T, F], preds_global [sumP, C] in GLOBAL tile order,
                     pred_off_global [T+1]); elsewhere None.
    A ``dist.gather`` of the two sizes per rank, then grouped point-to-point transfers (``batch_isend_irecv``: on RCCL one
    ncclGroupStart / ncclGroupEnd of send / recv pairs = gatherv) of exactly-sized buffers: tile ids and prediction counts
    as int64, per-tile values as float64, predictions in their own dtype.  Backends: "nccl" = RCCL over xGMI, "gloo"."""
    import torch
    import torch.distributed as dist

    dev = fixed.device if device is None else device
    T_r, F = fixed.shape
    Cp = preds.shape[1]
    n_pr = int(preds.shape[0])
    meta = torch.tensor([T_r, n_pr], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world_size)] if rank == dst else None
    dist.gather(meta, metas, dst=dst)
    ids = torch.stack([torch.as_tensor(np.asarray(tile_index, dtype=np.int64)),
                       torch.as_tensor(np.asarray(pred_counts, dtype=np.int64))], dim=1).contiguous().to(dev)
    fx = fixed.to(torch.float64).contiguous()
    pr = preds.contiguous()
    if rank != dst:
        ops = []
        if T_r > 0:
            ops += [dist.P2POp(dist.isend, ids, dst), dist.P2POp(dist.isend, fx, dst)]
        if n_pr > 0:
            ops.append(dist.P2POp(dist.isend, pr, dst))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return None
    sizes = [(int(m[0].item()), int(m[1].item())) for m in metas]
    bufs, ops = {}, []
    for r, (tr, npr) in enumerate(sizes):
        if r == dst:
            bufs[r] = (ids, fx, pr)
            continue
        b = (torch.empty((tr, 2), dtype=torch.int64, device=dev), torch.empty((tr, F), dtype=torch.float64, device=dev),
             torch.empty((npr, Cp), dtype=preds.dtype, device=dev))
        bufs[r] = b
        if tr > 0:
            ops += [dist.P2POp(dist.irecv, b[0], r), dist.P2POp(dist.irecv, b[1], r)]
        if npr > 0:
            ops.append(dist.P2POp(dist.irecv, b[2], r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    shards = []
    for r in range(world_size):
        i_, f_, p_ = (t.cpu().numpy() for t in bufs[r])
        shards.append((f_, p_, i_[:, 1], i_[:, 0]))
    return assemble_global(shards, total)


def gather_arrays(fixed, preds, pred_counts, tile_index, total, world_size, rank, device_id=None, dst=0):
    """``gather_results`` for host arrays: tensors are staged on this rank's GPU when the process group runs on
    RCCL ("nccl"), on the host for gloo."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != world_size:
        raise RuntimeError(f"gather_arrays: no torch.distributed process group of {world_size} ranks is initialised")
    dev = torch.device("cpu")
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", int(device_id or 0))
    fx = torch.as_tensor(np.ascontiguousarray(fixed, dtype=np.float64), device=dev)
    pr = torch.as_tensor(np.ascontiguousarray(preds), device=dev)
    return gather_results(fx, pr, pred_counts, tile_index, world_size, rank, device=dev, dst=dst, total=total)


def run_sharded(engine, batch, world_size, rank=None, n_eval=20, **fit_kw):
    """Partition -> pack -> engine -> results of ONE global packed batch.

    batch: dict with D, obs_off, pred_off, X, y, Xs (host arrays, global tile order) and optionally theta0 / lo / hi
    [T, H].  With ``rank`` given, this rank's shard is run and (fixed, preds, counts, tile_index) returned for
    ``gather_results``; with ``rank=None`` all ``world_size`` LOGICAL shards are run one after the other on this
    engine and merged with ``assemble_global`` -- the single-GPU rehearsal of the multi-GPU path (BASELINE configs[3]).
    fixed columns: theta (H), nll, status, n_eval, n_iter."""
    obs_off, pred_off = np.asarray(batch["obs_off"]), np.asarray(batch["pred_off"])
    D = batch["D"]
    T = len(obs_off) - 1
    parts = partition_tiles(np.diff(obs_off), np.diff(pred_off), world_size, n_eval)

    def one(r):
        ids = parts[r]
        sub = pack_subset(batch, ids)
        kw = dict(fit_kw)
        for k in ("theta0", "lo", "hi"):
            if k in batch and batch[k] is not None:
                kw[k] = np.asarray(batch[k])[ids]
        res = engine.fit_predict_batch(D=D, obs_off=sub["obs_off"], X=sub["X"], y=sub["y"], pred_off=sub["pred_off"],
                                       Xs=sub["Xs"], **kw)
        n_iter = res.n_iter if res.n_iter is not None else np.zeros(len(ids))
        fixed = np.concatenate([res.theta, res.nll[:, None], res.status[:, None].astype(np.float64),
                                res.n_eval[:, None].astype(np.float64), np.asarray(n_iter, dtype=np.float64)[:, None]], axis=1)
        preds = np.stack([np.asarray(res.f_mean), np.asarray(res.f_var), np.asarray(res.y_var)], axis=1)
        return fixed, preds, np.diff(sub["pred_off"]), ids, res

    if rank is not None:
        return one(rank)
    shards = [one(r) for r in range(world_size)]
    fixed_g, preds_g, pred_off_g = assemble_global([s[:4] for s in shards], T)
    return fixed_g, preds_g, pred_off_g, [s[4] for s in shards]


def all_gather_equal(fixed, preds, world_size):
    """Equal-size variant used by the weak-scaling bench (every rank owns the same number of tiles and
    predictions): two all_gather calls, rank-major order = global tile order.  Returns (fixed_all, preds_all)."""
    import torch
    import torch.distributed as dist
    fl = [torch.empty_like(fixed) for _ in range(world_size)]
    dist.all_gather(fl, fixed)
    pl = [torch.empty_like(preds) for _ in range(world_size)]
    dist.all_gather(pl, preds)
    return torch.cat(fl, dim=0), torch.cat(pl, dim=0)
