#!/usr/bin/env python3
"""Full-size parity distribution: BASELINE configs[1]'s 4096 distinct tiles (the bench's seeds) through
  (a) the fp32 kernels with the bench's settings (max_iter 20, default tolerances),
  (b) the fp32 kernels run to convergence (max_iter 500),
  (c) the fp64 kernels run to convergence (max_iter 500; pinned to the oracle at 1e-9),
and the quantiles of |dl|/l, |dsf2|/sf2, |dsn2|/sn2, dNLL/N, |df*|/max|y| of (a) and (b) against (c), plus the histogram of
evaluations per tile.  Prints one JSON object.  usage: scripts/parity_distribution.py [--tiles 4096] [--out file.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def quant(v):
    v = np.asarray(v, dtype=np.float64)
    v = v[np.isfinite(v)]
    return {"median": float(np.median(v)), "p90": float(np.quantile(v, 0.9)), "p99": float(np.quantile(v, 0.99)),
            "max": float(v.max()), "min": float(v.min())}


def compare(r, ref, Ns, ymax, D, P):
    T = len(Ns)
    out = {"lengthscale_rel": quant(np.abs(r.theta[:, :D] - ref.theta[:, :D]) / ref.theta[:, :D]),
           "kernel_variance_rel": quant(np.abs(r.theta[:, D] - ref.theta[:, D]) / ref.theta[:, D]),
           "likelihood_variance_rel": quant(np.abs(r.theta[:, D + 1] - ref.theta[:, D + 1]) / ref.theta[:, D + 1]),
           "nll_gap_per_obs": quant((r.nll - ref.nll) / Ns),
           "f_mean_abs_over_ymax": quant(np.max(np.abs(np.asarray(r.f_mean, np.float64).reshape(T, P) -
                                                       np.asarray(ref.f_mean, np.float64).reshape(T, P)), axis=1) / ymax),
           "f_var_abs_over_sf2": quant(np.max(np.abs(np.asarray(r.f_var, np.float64).reshape(T, P) -
                                                     np.asarray(ref.f_var, np.float64).reshape(T, P)), axis=1) / ref.theta[:, D])}
    # length scales pinned at their upper bound (a flat direction) are compared at the bound on both sides
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=4096)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import bench
    ns = argparse.Namespace(tiles=a.tiles, nobs=500, npred=500, dim=3, kernel="RBF", optimiser="lbfgs", max_iter=20,
                            exact_iters=False, global_tiles=0, workload="configs1")
    w = bench.build_workload(ns, 0, 1, 16)                       # forks its workers BEFORE the GPU is initialised
    from gpsat_amd.engine import Engine
    eng = Engine(0)
    T, D, P = w["T"], w["D"], w["P"]
    kw = dict(D=D, obs_off=w["obs_off"], pred_off=w["pred_off"], theta0=w["theta0"], lo=w["lo"], hi=w["hi"], kernel="RBF",
              optimiser="lbfgs")
    X64, y64, Xs64 = (np.asarray(w[k], dtype=np.float64) for k in ("X", "y", "Xs"))
    r_a = eng.fit_predict_batch(X=w["X"], y=w["y"], Xs=w["Xs"], max_iter=20, **kw)
    r_b = eng.fit_predict_batch(X=w["X"], y=w["y"], Xs=w["Xs"], max_iter=500, **kw)
    r_c = eng.fit_predict_batch(X=X64, y=y64, Xs=Xs64, max_iter=500, dtype="f64", **kw)
    r_d = eng.fit_predict_batch(X=X64, y=y64, Xs=Xs64, max_iter=20, dtype="f64", **kw)
    Ns = np.diff(w["obs_off"]).astype(np.float64)
    ymax = np.array([np.abs(y64[w["obs_off"][t]:w["obs_off"][t + 1]]).max() for t in range(T)])
    res = {"tiles": T,
           "fp32_bench_settings_vs_fp64_converged": compare(r_a, r_c, Ns, ymax, D, P),
           "fp32_converged_vs_fp64_converged": compare(r_b, r_c, Ns, ymax, D, P),
           "fp64_bench_settings_vs_fp64_converged": compare(r_d, r_c, Ns, ymax, D, P),
           "evals": {k: {"mean": float(r.n_eval.mean()), "p50": float(np.median(r.n_eval)), "p90": float(np.quantile(r.n_eval, 0.9)),
                         "p99": float(np.quantile(r.n_eval, 0.99)), "max": int(r.n_eval.max()),
                         "iters_mean": float(r.n_iter.mean()),
                         "status": {str(s): int((r.status == s).sum()) for s in np.unique(r.status)},
                         "hist": np.bincount(np.minimum(r.n_eval, 99) // 5, minlength=20).tolist()}
                     for k, r in (("fp32_bench", r_a), ("fp32_converged", r_b), ("fp64_converged", r_c), ("fp64_bench", r_d))},
           "kernel_ms": {"fp32_bench": r_a.kernel_ms, "fp32_converged": r_b.kernel_ms, "fp64_converged": r_c.kernel_ms,
                         "fp64_bench": r_d.kernel_ms}}
    # tiles whose fp64 optimum has a length scale at the upper bound of its box: a flat direction of the objective
    hi_l = w["hi"][:, :D]
    res["fp64_lengthscales_at_bound_frac"] = float(np.mean((r_c.theta[:, :D] > 0.98 * hi_l).any(axis=1)))
    s = json.dumps(res)
    print(s)
    if a.out:
        open(a.out, "w").write(s)
    eng.close()


if __name__ == "__main__":
    main()
