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


def gather_results(fixed, preds, pred_counts, tile_index, world_size, rank, device=None, dst=0):
    """One gather of per-tile results to ``dst``.

    fixed       : [T_r, F] float32 torch tensor (theta, nll, status, n_eval ... per local tile)
    preds       : [sumP_r, 3] float32 torch tensor (f*, f*_var, y_var)
    pred_counts : [T_r] int64 numpy, predictions per local tile
    tile_index  : [T_r] int64 numpy, global tile ids of the local tiles
    Returns on dst: (fixed_global [T, F], preds_global [sumP, 3] in GLOBAL tile order,
                     pred_off_global [T+1]); elsewhere None.
    Variable-length parts are padded to the maximum over ranks and moved by ONE all_gather each
    (payloads are tiny against 7 x ~153 GB/s xGMI links; ordering correctness is what matters).
    """
    import torch
    import torch.distributed as dist

    dev = fixed.device if device is None else device
    T_r, F = fixed.shape
    meta = torch.tensor([T_r, preds.shape[0]], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world_size)]
    dist.all_gather(metas, meta)
    Tmax = int(max(m[0].item() for m in metas))
    Pmax = int(max(m[1].item() for m in metas))
    fx = torch.zeros((Tmax, F + 2), dtype=torch.float64, device=dev)
    fx[:T_r, :F] = fixed.to(torch.float64)
    fx[:T_r, F] = torch.as_tensor(tile_index, dtype=torch.float64, device=dev)
    fx[:T_r, F + 1] = torch.as_tensor(pred_counts, dtype=torch.float64, device=dev)
    pr = torch.zeros((Pmax, 3), dtype=torch.float32, device=dev)
    pr[:preds.shape[0]] = preds
    fxs = [torch.zeros_like(fx) for _ in range(world_size)]
    prs = [torch.zeros_like(pr) for _ in range(world_size)]
    dist.all_gather(fxs, fx)
    dist.all_gather(prs, pr)
    if rank != dst:
        return None
    Ttot = int(sum(m[0].item() for m in metas))
    fixed_g = np.zeros((Ttot, F))
    counts_g = np.zeros(Ttot, dtype=np.int64)
    chunks = {}
    for r in range(world_size):
        tr = int(metas[r][0].item())
        f = fxs[r][:tr].cpu().numpy()
        ids = f[:, F].astype(np.int64)
        cnt = f[:, F + 1].astype(np.int64)
        fixed_g[ids] = f[:, :F]
        counts_g[ids] = cnt
        off = np.concatenate([[0], np.cumsum(cnt)])
        p = prs[r].cpu().numpy()
        for k, t in enumerate(ids):
            chunks[int(t)] = p[off[k]:off[k + 1]]
    pred_off_g = np.concatenate([[0], np.cumsum(counts_g)])
    preds_g = np.concatenate([chunks[t] for t in range(Ttot)], axis=0) if Ttot else np.zeros((0, 3), np.float32)
    return fixed_g, preds_g, pred_off_g


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
