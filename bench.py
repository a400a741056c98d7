#!/usr/bin/env python3
"""bench.py -- local-expert tiles/sec (fit + predict), the BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic expert tiles resident in HBM:
for every tile, L-BFGS fit of the 5 hyper-parameters (max_iter optimiser iterations), the objective
at the optimum, and the predictive mean / variances at P points -- all inside ONE persistent
gfx950 kernel launch per step (gpsat_fit_predict_batch).  Workload at N=1 = BASELINE.json
configs[1]: 4,096 tiles, 500 obs/tile, RBF, 3-D inputs, fp32, 20 optimiser steps, P = 500.
Tiles shard embarrassingly: every rank owns --tiles tiles (weak scaling), no data-path collective;
one RCCL all_gather of per-tile hyper-parameters + predictions closes each step when N > 1.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table


def f_eval(N, D):     # one objective+gradient evaluation
    return N ** 3 + (3.5 * D + 9) * N * N


def f_nll(N, D):      # objective only (final factorisation when it is not reused)
    return N ** 3 / 3 + N * N + (3 * D + 6) * N * N / 2


def f_pred(N, P, D):
    return N * N * P + 2 * N * P + (3 * D + 6) * N * P / 2 + P * N


def measured_traffic(a, T, N, P, D):
    """HBM bytes per launch measured with rocprofv3 PMC passes (profiles/traffic.json), only when the
    workload is the one it was measured on; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        w = t["workload"]
        same = (w["tiles_per_gpu"], w["obs_per_tile"], w["pred_per_tile"], w["dim"], w["kernel"], w["optimiser"],
                w["max_iter"]) == (T, N, P, D, a.kernel, a.optimiser, a.max_iter)
        return float(t["hbm_bytes_per_launch"]) if same else None
    except Exception:
        return None


def _gen_tile(args):
    from gpsat_amd import synthetic as syn
    seed, N, P, D, kid = args
    return syn.make_tile(seed, N, P, D, kid)


def make_tiles(T, N, P, D, kid, base_seed, workers):
    """Synthetic tiles, generated in parallel on the host."""
    from multiprocessing import get_context
    jobs = [(base_seed + t, N, P, D, kid) for t in range(T)]
    if workers > 1:
        with get_context("fork").Pool(workers) as pool:
            res = pool.map(_gen_tile, jobs, chunksize=max(1, T // (workers * 8)))
    else:
        res = [_gen_tile(j) for j in jobs]
    X = np.concatenate([r[0] for r in res]).astype(np.float32)
    y = np.concatenate([r[1] for r in res]).astype(np.float32)
    Xs = np.concatenate([r[2] for r in res]).astype(np.float32)
    obs_off = np.arange(T + 1, dtype=np.int64) * N
    pred_off = np.arange(T + 1, dtype=np.int64) * P
    return X, y, Xs, obs_off, pred_off


def _cpu_tile(args):
    """cpu_baseline leg: the fp64 oracle (a port of the reference's algorithm) on one tile."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(1)
    except Exception:
        ctx = None
    from oracle import gp_oracle as go
    X, y, Xs, D, kid, max_iter, lo, hi = args
    t0 = time.perf_counter()
    o = go.fit_predict_batch(kid, D, np.array([0, len(y)]), X, y, np.array([0, len(Xs)]), Xs,
                             np.ones((1, D + 2)), lo[None], hi[None], np.ones(D + 2, bool), max_iter=max_iter)
    dt = time.perf_counter() - t0
    return dt, int(o["n_eval"][0])


def cpu_baseline(X, y, Xs, N, P, D, kid, max_iter, n_tiles, workers):
    from multiprocessing import get_context
    from gpsat_amd import synthetic as syn
    lo, hi = syn.default_bounds(1, D)
    jobs = [(X[t * N:(t + 1) * N].astype(np.float64), y[t * N:(t + 1) * N].astype(np.float64),
             Xs[t * P:(t + 1) * P].astype(np.float64), D, kid, max_iter, lo[0], hi[0]) for t in range(n_tiles)]
    t0 = time.perf_counter()
    with get_context("fork").Pool(workers) as pool:
        res = pool.map(_cpu_tile, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    return n_tiles / wall, float(np.mean([r[1] for r in res])), wall


def other_workload(a):
    """The other single-GPU shapes of BASELINE.json as non-default bench lines (same timing protocol; prototype tiles
    per size class are generated once and replicated -- the arithmetic does not depend on the values)."""
    from gpsat_amd import synthetic as syn
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    D, P = 3, 500
    if a.workload == "configs2":
        T = a.tiles if a.tiles != 4096 else 1024
        kid, kernel, dtype, np_dt, optimiser, max_iter = 2, "Matern32", "f32", np.float32, "lbfgs", a.max_iter
        sizes = [128, 256, 384, 512, 768, 1024, 1536, 2048]
        Ns = np.random.default_rng(rank).choice(sizes, T)
        proto = {n: [syn.make_tile(100 + 10 * i + j, n, P, D, kid) for j in range(2)] for i, n in enumerate(sizes)}
        parts = [proto[int(n)][t % 2] for t, n in enumerate(Ns)]
        theta0 = np.ones((T, D + 2))
        lo, hi = syn.default_bounds(T, D)
        name = "BASELINE.json configs[2]: ragged tiles N in {128..2048}, Matern-3/2, 3D inputs, fp32"
        peak = PEAK_F32_MFMA_TFLOPS
    else:
        T = a.tiles if a.tiles != 4096 else 1024
        kid, kernel, dtype, np_dt, optimiser, max_iter = 0, "RBF", "f64", np.float64, "none", 0
        Ns = np.full(T, 2000)
        proto = [syn.make_tile(1 + j, 2000, P, D, kid) for j in range(8)]
        parts = [proto[t % 8] for t in range(T)]
        theta0 = np.stack([pp[3] for pp in parts])          # the generating hyper-parameters stand in for the smoothed ones
        lo = hi = None
        name = "BASELINE.json configs[4] per GPU: fp64, N=2000, predict-only with given hyper-parameters"
        peak = 78.6                                         # fp64 MFMA, MI355X_MICROARCH.md
    X = np.concatenate([pp[0] for pp in parts]).astype(np_dt)
    y = np.concatenate([pp[1] for pp in parts]).astype(np_dt)
    Xs = np.concatenate([pp[2] for pp in parts]).astype(np_dt)
    obs_off = np.concatenate([[0], np.cumsum(Ns)])
    pred_off = np.arange(T + 1) * P

    import torch
    import torch.distributed as dist
    from gpsat_amd.engine import Engine
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
    eng = Engine(local_rank)
    kw = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=pred_off, Xs=dXs, theta0=theta0, kernel=kernel, optimiser=optimiser,
              max_iter=max_iter, dtype=dtype)
    if lo is not None:
        kw.update(lo=lo, hi=hi)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        eng.fit_predict_batch(**kw)
    barrier()
    t0 = time.perf_counter()
    kms = []
    for _ in range(a.steps):
        r = eng.fit_predict_batch(**kw)
        kms.append(r.kernel_ms)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        Nf = Ns.astype(np.float64)
        if a.workload == "configs2":
            flops = float((r.n_eval * f_eval(Nf, D) + f_pred(Nf, P, D)).sum())
        else:
            flops = float((f_nll(Nf, D) + f_pred(Nf, P, D)).sum())
        k_ms = float(np.mean(kms))
        ach = flops / (k_ms * 1e-3) / 1e12
        print(json.dumps({
            "metric": "local-expert tiles/sec (fit+predict)", "value": round(T * world * a.steps / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic (prototype tiles per size class, replicated)",
            "config": {"workload": name, "tiles_per_gpu": int(T), "mean_obs_per_tile": float(Nf.mean()), "pred_per_tile": P,
                       "dim": D, "kernel": kernel, "optimiser": optimiser, "max_iter": max_iter,
                       "evals_per_tile": round(float(r.n_eval.mean()), 2), "failed_tiles": int(np.sum((r.status == 2) | (r.status == 3))),
                       "parallelism": f"tile-sharded x{world}", "device": eng.device_name},
            "roofline": {"bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": None, "kernel_ms": round(k_ms, 3), "flops_per_launch": flops},
        }), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=4096, help="tiles per GPU (weak scaling)")
    ap.add_argument("--nobs", type=int, default=500)
    ap.add_argument("--npred", type=int, default=500)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--kernel", default="RBF")
    ap.add_argument("--optimiser", default="lbfgs")
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--cpu-tiles", type=int, default=-1, help="tiles in the CPU baseline sample (0 = skip)")
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--workers", type=int, default=0, help="host processes for data generation / CPU baseline "
                    "(0 = auto; use 1 under rocprofv3 --pmc: no fork beside the profiler)")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2", "configs4"],
                    help="BASELINE.json configs[i]: 1 = the metric's workload (default); 2 = ragged N in {128..2048}, "
                         "Matern-3/2, fp32; 4 = fp64, N = 2000, predict-only with given hyper-parameters")
    a = ap.parse_args()
    if a.workload != "configs1":
        return other_workload(a)

    from gpsat_amd import _lib as L
    from gpsat_amd import synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")

    T, N, P, D = a.tiles, a.nobs, a.npred, a.dim
    kid = L.KERNEL_IDS[a.kernel]
    H = D + 2
    ncpu = os.cpu_count() or 1
    workers = a.workers if a.workers > 0 else max(1, min(16, ncpu // max(1, min(world, 8))))
    # ---- host-side work that forks worker processes happens BEFORE the GPU / RCCL are initialised
    X, y, Xs, obs_off, pred_off = make_tiles(T, N, P, D, kid, base_seed=1_000_000 * rank, workers=workers)
    cpu = None
    n_cpu_tiles = a.cpu_tiles if a.cpu_tiles >= 0 else (2 * workers if world == 1 else 0)
    if rank == 0 and n_cpu_tiles > 0:
        v, e_cpu, wall = cpu_baseline(X, y, Xs, N, P, D, kid, a.max_iter, n_cpu_tiles, workers)
        cpu = {"value": round(v, 3), "unit": "tiles/s", "cores": workers, "kind": "port",
               "sample": f"{n_cpu_tiles} of the same tiles, fp64 NumPy/SciPy oracle "
                         f"(L-BFGS-B maxiter={a.max_iter}, {e_cpu:.1f} evals/tile, predict P={P}), "
                         f"one single-threaded process per core, {wall:.1f}s wall"}

    import torch
    import torch.distributed as dist
    from gpsat_amd.engine import Engine
    from gpsat_amd.sharding import all_gather_equal

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
    fm = torch.empty(T * P, dtype=torch.float32, device=dev)
    fv = torch.empty_like(fm)
    yv = torch.empty_like(fm)
    theta0 = np.ones((T, H))
    lo, hi = syn.default_bounds(T, D)
    eng = Engine(local_rank, workgroups_per_cu=a.wg_per_cu)

    def step():
        r = eng.fit_predict_batch(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=pred_off, Xs=dXs, theta0=theta0,
                                  lo=lo, hi=hi, kernel=a.kernel, optimiser=a.optimiser, max_iter=a.max_iter,
                                  out=(fm, fv, yv))
        if use_dist:
            # final gather of per-tile hyper-parameters + predictions (the only collective of the path)
            fixed = torch.from_numpy(np.concatenate([r.theta, r.nll[:, None], r.status[:, None].astype(np.float64),
                                                     r.n_eval[:, None].astype(np.float64)], axis=1)).to(dev)
            all_gather_equal(fixed, torch.stack([fm, fv, yv], dim=1), world)
        return r

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, n_eval_sum, statuses = [], 0, None
    for _ in range(a.steps):
        r = step()
        kernel_ms.append(r.kernel_ms)
        n_eval_sum = int(r.n_eval.sum())
        statuses = r.status
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        tiles_total = T * world * a.steps
        value = tiles_total / dt
        E = n_eval_sum / T                                    # evaluations per tile actually performed
        flops_launch = T * (E * f_eval(N, D) + f_pred(N, P, D))
        k_ms = float(np.mean(kernel_ms))
        achieved = flops_launch / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "local-expert tiles/sec (fit+predict)", "value": round(value, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: synthetic tiles, RBF, 3D inputs, fp32",
                       "tiles_per_gpu": T, "obs_per_tile": N, "pred_per_tile": P, "dim": D, "kernel": a.kernel,
                       "optimiser": a.optimiser, "max_iter": a.max_iter, "evals_per_tile": round(E, 2),
                       "converged_frac": round(float(np.mean(statuses == 0)), 3),
                       "failed_tiles": int(np.sum(statuses >= 2)),
                       "parallelism": f"tile-sharded x{world}", "device": eng.device_name},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": measured_traffic(a, T, N, P, D),
                         "kernel": f"gp_tile_kernel<{D}, {kid}>", "kernel_ms": round(k_ms, 3),
                         "flops_per_launch": flops_launch},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
