#!/usr/bin/env python3
"""bench.py -- local-expert tiles/sec (fit + predict), the BASELINE.json metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic expert tiles resident in HBM: for every tile, L-BFGS
fit of the 5 hyper-parameters (max_iter optimiser iterations), the objective at the optimum, and the predictive mean /
variances at P points -- all inside ONE persistent gfx950 kernel launch per step (gpsat_fit_predict_batch).

Workloads (``--workload``):
  configs1 (default)  BASELINE.json configs[1], the metric's workload: 4,096 tiles x 500 obs, RBF, 3-D, fp32, 20
                      optimiser steps, P = 500.  Weak scaling: every rank owns --tiles tiles, no data-path collective,
                      one RCCL all_gather of per-tile hyper-parameters + predictions closes each step when N > 1.
  configs2            BASELINE.json configs[2]: ragged N in {128..2048}, Matern-3/2, fp32.
  configs4            BASELINE.json configs[4] per GPU: fp64, N = 2000, predict-only with given hyper-parameters.
  f64fit              configs[1]'s tiles in the reference's native precision (fp64 kernels, fit + predict).
  --global-tiles G    BASELINE.json configs[3]: ONE global list of G tiles (N = 500, RBF), split over the ranks by the
                      LPT partition (sharding.partition_tiles), each rank runs its shard, one gather(v) of
                      hyper-parameters + predictions to rank 0 in the reference's tile order.  Strong scaling.
``--exact-iters`` switches the optimiser's early stopping off (ftol = gtol = off): every tile runs exactly max_iter
iterations unless its line search fails; the CPU baseline then runs SciPy with ftol = gtol = 0 on matched work.

``python bench.py --gpus N`` without a torch.distributed environment starts the N ranks itself (a child
``python -m torch.distributed.run --nproc-per-node N ... bench.py``, before this process touches the GPU) and relays the line.

Prints ONE JSON line on rank 0.  ``value`` = whole-job tiles/s with inputs resident in HBM; ``host_to_host_ms`` is the
same step timed from packed host arrays to outputs on the host (SURVEY.md 8(d)'s definition, PCIe included) -- context,
never ``value``.
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
PEAK_F64_MFMA_TFLOPS = 78.6


def f_eval(N, D):     # one objective+gradient evaluation
    return N ** 3 + (3.5 * D + 9) * N * N


def f_nll(N, D):      # objective only (final factorisation when it is not reused)
    return N ** 3 / 3 + N * N + (3 * D + 6) * N * N / 2


def f_pred(N, P, D):
    return N * N * P + 2 * N * P + (3 * D + 6) * N * P / 2 + P * N


def measured_traffic(key, sig):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE doubled per
    the gfx950 rule + WRITE_SIZE, separate --pmc passes of this same command), only when the workload signature
    matches the one it was measured on; otherwise null."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = t[key] if key in t else (t if key == "configs1" and "workload" in t else None)
        if e is None:
            return None, None
        w = e["workload"]
        same = all(sig.get(k) == v for k, v in w.items())
        return (float(e["hbm_bytes_per_launch"]), e.get("source")) if same else (None, None)
    except Exception:
        return None, None


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (cpu.max).  The GPU box gives
    one GPU's share of the host: 256 hardware threads visible, a quota of 16 CPUs -- more processes than that only
    time-share the quota (measured: 128 processes 18 tiles/s against 41.7 with 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if "model name" in line:
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ----------------------------------------------------------------------------------------------------------
# synthetic tiles, generated in parallel on the host BEFORE the GPU is initialised
# ----------------------------------------------------------------------------------------------------------
def _gen_tile(args):
    from gpsat_amd import synthetic as syn
    seed, N, P, D, kid = args
    return syn.make_tile(seed, N, P, D, kid)


def gen_tiles(jobs, workers):
    from multiprocessing import get_context
    if workers > 1 and len(jobs) > 1:
        with get_context("fork").Pool(workers) as pool:
            return pool.map(_gen_tile, jobs, chunksize=max(1, len(jobs) // (workers * 8)))
    return [_gen_tile(j) for j in jobs]


def build_workload(a, rank, world, workers):
    """Returns the rank's packed batch + metadata.  Prototype tiles are generated once and replicated where the tile
    count is large (the arithmetic does not depend on the values; replicas are made distinct by scaling y)."""
    from gpsat_amd import _lib as L
    from gpsat_amd import synthetic as syn
    D, P = a.dim, a.npred
    w = dict(D=D, P=P, peak=PEAK_F32_MFMA_TFLOPS, dtype="f32", np_dt=np.float32, scaling="weak", global_T=None)
    if a.global_tiles > 0:
        G, N, kid = a.global_tiles, a.nobs, L.KERNEL_IDS[a.kernel]
        from gpsat_amd import sharding
        parts = sharding.partition_tiles(np.full(G, N), np.full(G, P), world, a.max_iter)
        mine = parts[rank]
        NPROTO = min(G, 2048)
        proto = gen_tiles([(9_000_000 + j, N, P, D, kid) for j in range(NPROTO)], workers)
        pX = np.stack([p[0] for p in proto]).astype(np.float32)
        py = np.stack([p[1] for p in proto])
        pXs = np.stack([p[2] for p in proto]).astype(np.float32)
        j, rep = mine % NPROTO, mine // NPROTO
        X = pX[j].reshape(-1, D)
        y = (py[j] * (1.0 + 0.01 * rep)[:, None]).reshape(-1).astype(np.float32)
        Xs = pXs[j].reshape(-1, D)
        T = len(mine)
        lo, hi = syn.default_bounds(T, D)
        w.update(name=f"BASELINE.json configs[3]: ONE list of {G} synthetic tiles, N={N}, RBF, 3D inputs, fp32, LPT-sharded",
                 key="configs3", T=T, Ns=np.full(T, N), kid=kid, kernel=a.kernel, optimiser=a.optimiser, max_iter=a.max_iter,
                 X=X, y=y, Xs=Xs, theta0=np.ones((T, D + 2)), lo=lo, hi=hi, scaling="strong", global_T=G, mine=mine,
                 data="synthetic (2048 prototype tiles, replicas distinct by scaled observations)")
    elif a.workload in ("configs1", "f64fit"):
        T, N, kid = a.tiles, a.nobs, L.KERNEL_IDS[a.kernel]
        res = gen_tiles([(1_000_000 * rank + t, N, P, D, kid) for t in range(T)], workers)
        lo, hi = syn.default_bounds(T, D)
        w.update(name="BASELINE.json configs[1]: synthetic tiles, RBF, 3D inputs, fp32", key="configs1", T=T, Ns=np.full(T, N),
                 kid=kid, kernel=a.kernel, optimiser=a.optimiser, max_iter=a.max_iter,
                 X=np.concatenate([r[0] for r in res]).astype(np.float32), y=np.concatenate([r[1] for r in res]).astype(np.float32),
                 Xs=np.concatenate([r[2] for r in res]).astype(np.float32), theta0=np.ones((T, D + 2)), lo=lo, hi=hi,
                 data="synthetic")
        if a.workload == "f64fit":
            # the same tiles (fp32-representable coordinates) through the fp64 kernels: the reference's native precision
            w.update(name="BASELINE.json configs[1]'s tiles in fp64 (GPflow default_float): RBF, 3D inputs, fit + predict",
                     key="f64fit", dtype="f64", np_dt=np.float64, peak=PEAK_F64_MFMA_TFLOPS,
                     X=w["X"].astype(np.float64), y=w["y"].astype(np.float64), Xs=w["Xs"].astype(np.float64))
    elif a.workload == "configs2":
        T = a.tiles                     # BASELINE configs[2]: 4096 ragged tiles (a 1024-tile launch is as long as its largest tile)
        kid = 2
        sizes = [128, 256, 384, 512, 768, 1024, 1536, 2048]
        Ns = np.random.default_rng(rank).choice(sizes, T)
        NPR = 8                          # prototype tiles per size class (r2: two -- 16 optimiser trajectories in all made
        #                                  evaluations per tile jump by 15 % with any change of rounding)
        protos = gen_tiles([(100 + 10 * i + j, n, P, D, kid) for i, n in enumerate(sizes) for j in range(NPR)], workers)
        proto = {n: protos[NPR * i:NPR * i + NPR] for i, n in enumerate(sizes)}
        parts = [proto[int(n)][t % NPR] for t, n in enumerate(Ns)]
        lo, hi = syn.default_bounds(T, D)
        w.update(name="BASELINE.json configs[2]: ragged tiles N in {128..2048}, Matern-3/2, 3D inputs, fp32", key="configs2",
                 T=T, Ns=Ns, kid=kid, kernel="Matern32", optimiser="lbfgs", max_iter=a.max_iter,
                 X=np.concatenate([p[0] for p in parts]).astype(np.float32), y=np.concatenate([p[1] for p in parts]).astype(np.float32),
                 Xs=np.concatenate([p[2] for p in parts]).astype(np.float32), theta0=np.ones((T, D + 2)), lo=lo, hi=hi,
                 data="synthetic (eight prototype tiles per size class, replicated)")
    else:
        T = a.tiles if a.tiles != 4096 else 1024
        kid = 0
        protos = gen_tiles([(1 + j, 2000, P, D, kid) for j in range(8)], workers)
        parts = [protos[t % 8] for t in range(T)]
        w.update(name="BASELINE.json configs[4] per GPU: fp64, N=2000, predict-only with given hyper-parameters", key="configs4",
                 T=T, Ns=np.full(T, 2000), kid=kid, kernel="RBF", optimiser="none", max_iter=0, dtype="f64", np_dt=np.float64,
                 peak=PEAK_F64_MFMA_TFLOPS,
                 X=np.concatenate([p[0] for p in parts]), y=np.concatenate([p[1] for p in parts]),
                 Xs=np.concatenate([p[2] for p in parts]),
                 theta0=np.stack([p[3] for p in parts]),      # the generating parameters stand in for the smoothed ones
                 lo=None, hi=None, data="synthetic (eight prototype tiles, replicated)")
    w["obs_off"] = np.concatenate([[0], np.cumsum(w["Ns"])]).astype(np.int64)
    w["pred_off"] = np.arange(w["T"] + 1, dtype=np.int64) * P
    return w


# ----------------------------------------------------------------------------------------------------------
# cpu_baseline leg: the fp64 oracle (a port of the reference's algorithm) on a bounded sample of the same tiles
# ----------------------------------------------------------------------------------------------------------
def _cpu_warm(_):
    import scipy.linalg  # noqa: F401
    from oracle import gp_oracle  # noqa: F401
    return 0


def _cpu_tile(args):
    X, y, Xs, D, kid, max_iter, lo, hi, theta0, optimise, exact, nthreads = args
    from threadpoolctl import threadpool_limits
    from oracle import gp_oracle as go
    with threadpool_limits(nthreads):
        t0 = time.perf_counter()
        m = go.OracleGPR(X, y, kernel={0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}[kid])
        m.theta = np.array(theta0, dtype=np.float64)
        if lo is not None:
            box = np.isfinite(lo) & np.isfinite(hi)
            m.lo, m.hi = np.where(box, lo, -np.inf), np.where(box, hi, np.inf)
            m.shift = np.where(box, 0.0, m.shift)
        nit = 0
        if optimise:
            # SciPy defaults (what the reference runs), or both stopping tests off for matched work with --exact-iters
            opts = dict(maxiter=max_iter, ftol=0.0, gtol=0.0) if exact else dict(maxiter=max_iter)
            D_ = X.shape[1]
            u_all = go.u_from_theta(m.theta, m.lo, m.hi, m.shift)
            cnt = [0]

            def fun(u):
                th = go.theta_from_u(u, m.lo, m.hi, m.shift)
                f, g = go.nll_and_grad(kid, X, y, th)
                cnt[0] += 1
                if not np.isfinite(f):
                    return 1e300, np.zeros(D_ + 2)
                return f, g * go.dtheta_du(th, m.lo, m.hi, m.shift)
            from scipy.optimize import minimize
            res = minimize(fun, u_all, jac=True, method="L-BFGS-B", options=opts)
            m.theta = go.theta_from_u(res.x, m.lo, m.hi, m.shift)
            m.n_eval, nit = cnt[0], int(res.nit)
        else:
            m.n_eval = 0
        nll = m.get_objective_function_value()
        if len(Xs):
            m.predict(Xs, apply_scale=False)
        return time.perf_counter() - t0, int(m.n_eval), nit, float(nll)


def _sk_tile(args):
    """Optional second CPU line: scikit-learn's GaussianProcessRegressor -- the oracle of the
    reference's own known-answer test (tests/test_localexperts.py:22-49) -- on the identical tile: ARD kernel of the same
    family + white noise, L-BFGS-B from the same start, no restarts, predict mean + std."""
    X, y, Xs, D, kid, max_iter, lo, hi, theta0, optimise, exact, nthreads = args
    from threadpoolctl import threadpool_limits
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel
    with threadpool_limits(nthreads):
        t0 = time.perf_counter()
        ls = np.asarray(theta0[:D], dtype=np.float64)
        hi_l = np.where(np.isfinite(hi[:D]), hi[:D], 1e5) if hi is not None else np.full(D, 1e5)
        bounds = np.stack([np.full(D, 1e-5), hi_l], axis=1)
        base = RBF(ls, bounds) if kid == 0 else Matern(ls, bounds, nu={1: 0.5, 2: 1.5, 3: 2.5}[kid])
        k = ConstantKernel(float(theta0[D]), (1e-6, 1e5)) * base + WhiteKernel(float(theta0[D + 1]), (1e-6, 1e5))
        gp = GaussianProcessRegressor(kernel=k, optimizer="fmin_l_bfgs_b" if optimise else None, n_restarts_optimizer=0)
        gp.fit(X, y)
        if len(Xs):
            gp.predict(Xs, return_std=True)
        return time.perf_counter() - t0, 0, 0, float(-gp.log_marginal_likelihood_value_)


def cpu_baseline(w, a, workers):
    """Bounded sample (about 10-30 s of CPU work in all).  Layouts: one single-threaded process per core, and one process
    with all-thread BLAS; modes: the GPU run's iteration budget, and SciPy run to its default convergence."""
    from multiprocessing import get_context
    D, P = w["D"], w["P"]
    optimise = w["optimiser"] != "none"

    def jobs(n, max_iter, exact, nthreads):
        out = []
        for t in range(min(n, w["T"])):
            o0, o1 = w["obs_off"][t], w["obs_off"][t + 1]
            out.append((w["X"][o0:o1].astype(np.float64), w["y"][o0:o1].astype(np.float64),
                        w["Xs"][t * P:(t + 1) * P].astype(np.float64), D, w["kid"], max_iter,
                        None if w["lo"] is None else w["lo"][t], None if w["hi"] is None else w["hi"][t], w["theta0"][t],
                        optimise, exact, nthreads))
        return out
    mean_n3 = float(np.mean(w["Ns"].astype(np.float64) ** 3))
    scale = (500.0 ** 3) / mean_n3                                        # sample sizes quoted for N = 500 tiles
    n_main = a.cpu_tiles if a.cpu_tiles > 0 else max(workers, int(256 * min(1.0, scale * (1.0 if optimise else 8.0))))
    modes = {}
    with get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_warm, range(workers * 2))                             # pool start-up and imports are not timed
        t0 = time.perf_counter()
        res = pool.map(_cpu_tile, jobs(n_main, w["max_iter"], a.exact_iters, 1), chunksize=1)
        wall = time.perf_counter() - t0
        n_done = len(res)
        modes["budget_per_core"] = dict(tiles_per_s=round(n_done / wall, 3), tiles=n_done, wall_s=round(wall, 2),
                                        evals_per_tile=round(float(np.mean([r[1] for r in res])), 2),
                                        iters_per_tile=round(float(np.mean([r[2] for r in res])), 2),
                                        layout=f"{workers} single-threaded processes")
        if optimise and not a.exact_iters:
            t0 = time.perf_counter()
            res_c = pool.map(_cpu_tile, jobs(max(workers, n_main // 4), 10_000, False, 1), chunksize=1)
            wall_c = time.perf_counter() - t0
            modes["converged_per_core"] = dict(tiles_per_s=round(len(res_c) / wall_c, 3), tiles=len(res_c), wall_s=round(wall_c, 2),
                                               evals_per_tile=round(float(np.mean([r[1] for r in res_c])), 2),
                                               iters_per_tile=round(float(np.mean([r[2] for r in res_c])), 2),
                                               layout=f"{workers} single-threaded processes, SciPy default convergence (maxiter 10000)")
    # every core this process may run on (the GPU box gives one GPU's share of the host; the judge asked for the all-core
    # figure beside the 16-process one) and the scikit-learn line of BASELINE.md section 2
    ncore = usable_cpus()
    big = min(ncore, 128)
    if big > workers and not a.exact_iters:
        with get_context("fork").Pool(big) as pool:
            pool.map(_cpu_warm, range(big * 2))
            t0 = time.perf_counter()
            res_a = pool.map(_cpu_tile, jobs(max(2 * big, n_main), w["max_iter"], False, 1), chunksize=1)
            wall_a = time.perf_counter() - t0
        modes["budget_all_cores"] = dict(tiles_per_s=round(len(res_a) / wall_a, 3), tiles=len(res_a), wall_s=round(wall_a, 2),
                                         evals_per_tile=round(float(np.mean([r[1] for r in res_a])), 2),
                                         layout=f"{big} single-threaded processes ({ncore} CPUs usable by this process, "
                                                f"{os.cpu_count()} hardware threads on the host)")
    if optimise and not a.exact_iters and not a.no_sklearn:
        with get_context("fork").Pool(workers) as pool:
            pool.map(_cpu_warm, range(workers * 2))
            t0 = time.perf_counter()
            res_s = pool.map(_sk_tile, jobs(max(workers, n_main // 8), w["max_iter"], False, 1), chunksize=1)
            wall_s = time.perf_counter() - t0
        modes["sklearn_per_core"] = dict(tiles_per_s=round(len(res_s) / wall_s, 3), tiles=len(res_s), wall_s=round(wall_s, 2),
                                         layout=f"{workers} single-threaded processes, sklearn GaussianProcessRegressor "
                                                f"(n_restarts_optimizer=0, run to its own convergence), fit + predict")
    # all-thread BLAS, one process (after the pool is gone: BLAS threads and fork do not mix)
    n_blas = max(2, n_main // 16)
    t0 = time.perf_counter()
    res_b = [_cpu_tile(j) for j in jobs(n_blas, w["max_iter"], a.exact_iters, None)]
    wall_b = time.perf_counter() - t0
    modes["budget_allthread_blas"] = dict(tiles_per_s=round(len(res_b) / wall_b, 3), tiles=len(res_b), wall_s=round(wall_b, 2),
                                          evals_per_tile=round(float(np.mean([r[1] for r in res_b])), 2),
                                          layout=f"one process, BLAS on all {os.cpu_count()} threads")
    best = max([k for k in ("budget_per_core", "budget_allthread_blas", "budget_all_cores") if k in modes],
               key=lambda k: modes[k]["tiles_per_s"])
    m = modes[best]
    cores_of = {"budget_per_core": workers, "budget_all_cores": big, "budget_allthread_blas": os.cpu_count()}
    what = (f"L-BFGS-B maxiter={w['max_iter']}" + (", ftol=gtol=0" if a.exact_iters else ", SciPy default tolerances")) if optimise \
        else "objective + predict only"
    cpu_nll = {"tiles": n_done, "nll": [r[3] for r in res]}        # objective reached within the iteration budget, per tile
    return {"value": m["tiles_per_s"], "unit": "tiles/s", "cores": cores_of[best],
            "kind": "port", "cpu_model": cpu_model(), "host_threads": os.cpu_count(), "usable_cpus": usable_cpus(),
            "sample": f"{m['tiles']} of the same tiles, fp64 NumPy/SciPy oracle ({what}, {m['evals_per_tile']} evals/tile, "
                      f"predict P={P}), {m['layout']}, {m['wall_s']} s wall (pool start-up excluded); fastest of the layouts in `modes`",
            "modes": modes, "_nll": cpu_nll}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=4096, help="tiles per GPU (weak scaling)")
    ap.add_argument("--global-tiles", type=int, default=0, help="ONE global tile list split over the ranks (strong scaling, configs[3])")
    ap.add_argument("--nobs", type=int, default=500)
    ap.add_argument("--npred", type=int, default=500)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--kernel", default="RBF")
    ap.add_argument("--optimiser", default="lbfgs")
    ap.add_argument("--max-iter", type=int, default=20)
    ap.add_argument("--exact-iters", action="store_true", help="early stopping off: exactly max_iter iterations per tile")
    ap.add_argument("--cpu-tiles", type=int, default=-1, help="tiles in the CPU baseline sample (0 = skip, -1 = auto)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-to-host timing")
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--workers", type=int, default=0, help="host processes for data generation / CPU baseline "
                    "(0 = auto; use 1 under rocprofv3 --pmc: no fork beside the profiler)")
    ap.add_argument("--workload", default="configs1", choices=["configs1", "configs2", "configs4", "f64fit"])
    ap.add_argument("--no-sklearn", action="store_true", help="skip the scikit-learn line of the CPU baseline")
    ap.add_argument("--no-quality", action="store_true", help="skip the result-quality block (fp64 converged reference run)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the two-step runs of configs[2], configs[4] per GPU and the "
                    "fp64 fit that follow the headline's timed region")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks as a CHILD torch.distributed.run (this process has not touched the
        # GPU and never will), relay its output and exit with its code
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    ncpu = os.cpu_count() or 1
    workers = a.workers if a.workers > 0 else max(1, min(16, usable_cpus(), ncpu // max(1, min(world, 8))))

    # ---- host-side work that forks worker processes happens BEFORE the GPU / RCCL are initialised
    w = build_workload(a, rank, world, workers)
    # the other single-GPU configurations of BASELINE.json, measured after the headline's timed region:
    # their tiles are generated here, before the GPU is initialised (the generators fork)
    others = {}
    if rank == 0 and world == 1 and a.workload == "configs1" and a.global_tiles == 0 and not a.no_other_workloads and not a.exact_iters:
        import copy
        for key, tiles in (("configs2", 4096), ("configs4", 1024), ("f64fit", a.tiles)):
            if key == "f64fit":
                ow = dict(w)
                ow.update(name="BASELINE.json configs[1]'s tiles in fp64 (GPflow default_float): RBF, 3D inputs, fit + predict",
                          key="f64fit", dtype="f64", np_dt=np.float64, peak=PEAK_F64_MFMA_TFLOPS)
            else:
                oa = copy.copy(a)
                oa.workload, oa.tiles = key, tiles
                ow = build_workload(oa, rank, world, workers)
            others[key] = ow
    cpu = None
    if rank == 0 and world == 1 and a.cpu_tiles != 0 and workers > 1:
        cpu = cpu_baseline(w, a, workers)

    import torch
    import torch.distributed as dist
    from gpsat_amd import sharding
    from gpsat_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    T, D, P = w["T"], w["D"], w["P"]
    t_dt = torch.float32 if w["dtype"] == "f32" else torch.float64
    dX, dy, dXs = (torch.from_numpy(np.ascontiguousarray(v, dtype=w["np_dt"])).to(dev) for v in (w["X"], w["y"], w["Xs"]))
    fm = torch.empty(max(T * P, 1), dtype=t_dt, device=dev)
    fv, yv = torch.empty_like(fm), torch.empty_like(fm)
    eng = Engine(local_rank, workgroups_per_cu=a.wg_per_cu)
    kw = dict(D=D, obs_off=w["obs_off"], pred_off=w["pred_off"], theta0=w["theta0"], kernel=w["kernel"], optimiser=w["optimiser"],
              max_iter=w["max_iter"], dtype=w["dtype"])
    if w["lo"] is not None:
        kw.update(lo=w["lo"], hi=w["hi"])
    if a.exact_iters:
        kw.update(ftol=-1.0, gtol=-1.0)

    def step():
        r = eng.fit_predict_batch(X=dX, y=dy, Xs=dXs, out=(fm, fv, yv), **kw)
        if use_dist:
            fixed = torch.from_numpy(np.concatenate([r.theta, r.nll[:, None], r.status[:, None].astype(np.float64),
                                                     r.n_eval[:, None].astype(np.float64), r.n_iter[:, None].astype(np.float64)],
                                                    axis=1)).to(dev)
            preds = torch.stack([fm[:T * P], fv[:T * P], yv[:T * P]], dim=1)
            if w["scaling"] == "strong":
                # the only exchange the path has: gather(v) of per-tile results into the reference's tile order on rank 0
                sharding.gather_results(fixed, preds, np.full(T, P), w["mine"], world, rank, device=dev, total=w["global_T"])
            else:
                sharding.all_gather_equal(fixed, preds, world)
        return r

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(a.steps):
        r = step()
        kernel_ms.append(r.kernel_ms)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- the same step from packed host arrays to outputs on the host; context, never `value`
    host_ms = None
    if rank == 0 and not a.no_host_leg:
        hX, hy, hXs = (np.ascontiguousarray(v, dtype=w["np_dt"]) for v in (w["X"], w["y"], w["Xs"]))
        eng.fit_predict_batch(X=hX, y=hy, Xs=hXs, **kw)
        th = time.perf_counter()
        eng.fit_predict_batch(X=hX, y=hy, Xs=hXs, **kw)
        host_ms = (time.perf_counter() - th) * 1e3

    # ---- what the early stop of the default tolerances costs: the same tiles run to convergence by the
    # fp64 kernels (pinned to the oracle at 1e-9); objective gap per observation of the timed run's result, and of the CPU
    # leg's maxiter-budget result on its sample.  Outside the timed region.
    quality = None
    if rank == 0 and world == 1 and not a.no_quality and w["optimiser"] != "none" and w["key"] in ("configs1", "f64fit"):
        try:                                              # outside the timed region: a failure here must not cost the headline line
            kw64 = dict(kw, dtype="f64", max_iter=500)
            kw64.pop("ftol", None), kw64.pop("gtol", None)
            r64 = eng.fit_predict_batch(X=np.asarray(w["X"], dtype=np.float64), y=np.asarray(w["y"], dtype=np.float64),
                                        Xs=np.asarray(w["Xs"], dtype=np.float64)[:0].reshape(0, D), **dict(kw64, pred_off=np.zeros(T + 1, np.int64)))
            Nn = w["Ns"].astype(np.float64)
            gap = (r.nll - r64.nll) / Nn
            okm = np.isfinite(gap)
            rel_l = np.abs(r.theta[:, :D] - r64.theta[:, :D]) / r64.theta[:, :D]
            quality = {"reference": "the same tiles by the fp64 HIP kernels run to convergence (L-BFGS, SciPy-default ftol / gtol, max_iter 500)",
                       "ref_evals_per_tile": round(float(r64.n_eval.mean()), 2), "ref_converged_frac": round(float(np.mean(r64.status == 0)), 4),
                       "nll_gap_per_obs": {"median": float(np.median(gap[okm])), "p99": float(np.quantile(gap[okm], 0.99)),
                                           "max": float(gap[okm].max()), "min": float(gap[okm].min())},
                       "lengthscale_rel_diff": {"median": float(np.median(rel_l)), "p99": float(np.quantile(rel_l, 0.99))}}
            if cpu is not None and "_nll" in cpu:
                nc = cpu["_nll"]["tiles"]
                gc = (np.asarray(cpu["_nll"]["nll"]) - r64.nll[:nc]) / Nn[:nc]
                quality["cpu_budget_nll_gap_per_obs"] = {"tiles": nc, "median": float(np.median(gc)), "p99": float(np.quantile(gc, 0.99)),
                                                         "max": float(gc.max())}
                gg = gap[:nc]
                quality["gpu_nll_gap_per_obs_same_tiles"] = {"median": float(np.median(gg)), "p99": float(np.quantile(gg, 0.99)),
                                                             "max": float(gg.max())}
        except Exception as e:                            # noqa: BLE001
            quality = {"error": f"{type(e).__name__}: {e}"}
    if cpu is not None:
        cpu.pop("_nll", None)

    other_out = {}
    for key, ow in others.items():
        try:                                              # a failure here must not cost the headline line, which is measured already
            # same protocol, two steps each: inputs resident in HBM, kernel time from the library's HIP events
            oT, oP = ow["T"], ow["P"]
            o_dt = torch.float32 if ow["dtype"] == "f32" else torch.float64
            oX, oy, oXs = (torch.from_numpy(np.ascontiguousarray(v, dtype=ow["np_dt"])).to(dev) for v in (ow["X"], ow["y"], ow["Xs"]))
            ofm = torch.empty(max(oT * oP, 1), dtype=o_dt, device=dev)
            ofv, oyv = torch.empty_like(ofm), torch.empty_like(ofm)
            okw = dict(D=ow["D"], obs_off=ow["obs_off"], pred_off=ow["pred_off"], theta0=ow["theta0"], kernel=ow["kernel"],
                       optimiser=ow["optimiser"], max_iter=ow["max_iter"], dtype=ow["dtype"])
            if ow["lo"] is not None:
                okw.update(lo=ow["lo"], hi=ow["hi"])
            eng.fit_predict_batch(X=oX, y=oy, Xs=oXs, out=(ofm, ofv, oyv), **okw)          # warm-up
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            oks = []
            for _ in range(2):
                orr = eng.fit_predict_batch(X=oX, y=oy, Xs=oXs, out=(ofm, ofv, oyv), **okw)
                oks.append(orr.kernel_ms)
            torch.cuda.synchronize()
            odt = time.perf_counter() - t1
            oN = ow["Ns"].astype(np.float64)
            if ow["optimiser"] == "none":
                ofl = float((f_nll(oN, ow["D"]) + f_pred(oN, oP, ow["D"])).sum())
            else:
                ofl = float((orr.n_eval.astype(np.float64) * f_eval(oN, ow["D"]) + f_pred(oN, oP, ow["D"])).sum())
            okm = float(np.mean(oks))
            other_out[key] = {"workload": ow["name"], "value": round(2 * oT / odt, 2), "unit": "tiles/s", "steps": 2, "dtype": ow["dtype"],
                              "ms_per_step": round(odt / 2 * 1e3, 3), "kernel_ms": round(okm, 3), "tiles": int(oT),
                              "obs_per_tile": float(oN.mean()), "evals_per_tile": round(float(orr.n_eval.mean()), 2),
                              "failed_tiles": int(np.sum((orr.status == 2) | (orr.status == 3))),
                              "roofline": {"bound": "mfma", "achieved": round(ofl / (okm * 1e-3) / 1e12, 3), "peak": ow["peak"], "unit": "TFLOP/s",
                                           "frac": round(ofl / (okm * 1e-3) / 1e12 / ow["peak"], 4)}}
            del oX, oy, oXs, ofm, ofv, oyv
        except Exception as e:                            # noqa: BLE001
            other_out[key] = {"workload": ow["name"], "error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        Nf = w["Ns"].astype(np.float64)
        n_eval = r.n_eval.astype(np.float64)
        if w["optimiser"] == "none":
            flops_launch = float((f_nll(Nf, D) + f_pred(Nf, P, D)).sum())
        else:
            flops_launch = float((n_eval * f_eval(Nf, D) + f_pred(Nf, P, D)).sum())
        k_ms = float(np.mean(kernel_ms))
        achieved = flops_launch / (k_ms * 1e-3) / 1e12
        total_tiles = (w["global_T"] if w["scaling"] == "strong" else T * world) * a.steps
        sig = dict(tiles_per_gpu=int(T), obs_per_tile=float(Nf.mean()), pred_per_tile=P, dim=D, kernel=w["kernel"],
                   optimiser=w["optimiser"], max_iter=w["max_iter"], exact_iters=bool(a.exact_iters))
        traffic, traffic_src = measured_traffic(w["key"], sig)
        st = r.status
        out = {
            "metric": "local-expert tiles/sec (fit+predict)", "value": round(total_tiles / dt, 2), "unit": "tiles/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": w["scaling"], "vs_baseline": None, "dtype": w["dtype"], "data": w["data"],
            "host_to_host_ms": None if host_ms is None else round(host_ms, 3),
            "config": {"workload": w["name"], "tiles_per_gpu": int(T), "obs_per_tile": float(Nf.mean()), "pred_per_tile": P,
                       "dim": D, "kernel": w["kernel"], "optimiser": w["optimiser"], "max_iter": w["max_iter"],
                       "early_stopping": "off (exactly max_iter iterations)" if a.exact_iters else "ftol/gtol defaults",
                       "evals_per_tile": round(float(n_eval.mean()), 2),
                       "iters_per_tile": round(float(r.n_iter.mean()), 2),
                       "converged_frac": round(float(np.mean(st == 0)), 3), "max_iter_frac": round(float(np.mean(st == 1)), 3),
                       "ls_failed_frac": round(float(np.mean(st == 6)), 4),
                       "failed_tiles": int(np.sum((st == 2) | (st == 3))),
                       "parallelism": f"tile-sharded x{world}" + (" (LPT over one global list, gather to rank 0)" if w["scaling"] == "strong" else ""),
                       "device": eng.device_name},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": w["peak"], "unit": "TFLOP/s",
                         "frac": round(achieved / w["peak"], 4), "traffic": traffic, "traffic_measured_in_run": False,
                         "traffic_source": traffic_src,
                         "kernel": f"gp_tile_kernel<{D}, {w['kid']}>" + (" (fp64)" if w["dtype"] == "f64" else ""),
                         "kernel_ms": round(k_ms, 3), "flops_per_launch": flops_launch},
        }
        if w["global_T"]:
            out["config"]["global_tiles"] = int(w["global_T"])
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if quality is not None:
            out["quality"] = quality
        if other_out:
            out["other_workloads"] = other_out
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
