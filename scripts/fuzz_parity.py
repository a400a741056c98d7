"""Developer: randomised parity sweep on the GPU box -- random ragged batches (sizes, dimensions, covariance functions,
precisions, 4-/8-wave builds, fixed parameters) against the oracle at fixed parameters: objective, gradient,
predictions and (sometimes) the full covariance.  Prints the worst normalised error per quantity; exits 1 on a violation
of the test-suite bounds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

names = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
wgs = [int(v) for v in os.environ.get("FUZZ_WG", "1,2").split(",")]
engines = {v: Engine(0, workgroups_per_cu=v) for v in wgs}
worst = {}
bad = 0
t_start = time.time()
n_cases = int(os.environ.get("CASES", 60))
for case in range(n_cases):
    D = int(rng.integers(1, 4)); kid = int(rng.integers(0, 4)); dtype = "f64" if rng.random() < 0.35 else "f32"
    T = int(rng.integers(1, 7))
    Ns = [int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 257, 300, 480, 511, 512, 513, 640])) for _ in range(T)]
    if rng.random() < 0.08:
        Ns[0] = int(rng.choice([900, 1024, 1300, 2048]))          # one workgroup per CU (LDS), long sweeps
    Ps = [int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 64, 65, 100, 130])) for _ in range(T)]
    wg = int(rng.choice(wgs))
    cov = rng.random() < 0.3
    np_dt = np.float32 if dtype == "f32" else np.float64
    b = syn.make_batch(T, Ns, Ps, D, kid, base_seed=int(rng.integers(0, 10**6)), dtype=np_dt)
    th0 = np.column_stack([rng.uniform(1.5, 6.0, (T, D)), rng.uniform(0.2, 1.5, T), rng.uniform(0.02, 0.3, T)])
    r = engines[wg].fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                                      theta0=th0, kernel=names[kid], optimiser="none", want_grad=True, dtype=dtype, full_cov=cov)
    for t in range(T):
        a, e = b["obs_off"][t], b["obs_off"][t + 1]; pa, pe = b["pred_off"][t], b["pred_off"][t + 1]
        N = e - a
        Xd, yd, Xsd = b["X"][a:e].astype(np.float64), b["y"][a:e].astype(np.float64), b["Xs"][pa:pe].astype(np.float64)
        nll, g = go.nll_and_grad(kid, Xd, yd, th0[t])
        f32 = dtype == "f32"
        # fp32: the error of the factorisation grows with cond(K) <= N sf2 / sn2 + 1; the suite's bound (2e-5 N) covers
        # cond ~ 5e4, beyond that the bound scales with it (seed 24, case 72: D = 1, Matern-5/2, N = 2048, l = 5.7,
        # sf2 / sn2 = 57: cond ~ 1.2e5, error 1.3 x the unscaled bound)
        kappa = max(1.0, (N * th0[t, D] / th0[t, D + 1]) / 5e4)
        errs = {"nll": abs(r.nll[t] - nll) / (((2e-5 * N + 2e-6 * abs(nll)) * kappa) if f32 else 1e-9 * max(1.0, abs(nll))),
                "grad": np.max(np.abs(r.grad[t] - g) / ((2e-3 if f32 else 1e-7) * (np.abs(g) + np.abs(g).max() + 1e-300)))}
        if pe > pa:
            f, fv, _ = go.predict(kid, Xd, yd, Xsd, th0[t])
            errs["f*"] = np.max(np.abs(np.asarray(r.f_mean[pa:pe], np.float64) - f)) / ((2e-3 if f32 else 1e-9) * max(np.abs(yd).max(), 1e-3))
            errs["f*_var"] = np.max(np.abs(np.asarray(r.f_var[pa:pe], np.float64) - fv)) / ((2e-3 if f32 else 1e-9) * th0[t, D] + (1e-6 if f32 else 0))
            if cov:
                C = np.asarray(r.f_cov[r.cov_off[t]:r.cov_off[t + 1]], np.float64).reshape(pe - pa, pe - pa)
                ref, _ = go.predict_cov(kid, Xd, yd, Xsd, th0[t])
                errs["f*_cov"] = np.max(np.abs(C - ref)) / ((4e-5 if f32 else 1e-9) * th0[t, D])
                if not np.array_equal(C, C.T):
                    errs["f*_cov"] = np.inf
        if r.status[t] != 5:
            errs["status"] = np.inf
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), float(v))
            if not v <= 1.0:
                bad += 1
                print(f"VIOLATION case {case} tile {t}: {k} = {v:.3g} x bound (D={D} k={names[kid]} {dtype} wg={wg} N={N} P={pe - pa})", flush=True)
    if case % 10 == 9:
        print(f"case {case + 1}/{n_cases}, {time.time() - t_start:.0f}s, worst (fraction of bound): " +
              ", ".join(f"{k} {v:.3f}" for k, v in worst.items()), flush=True)
print("violations:", bad)
sys.exit(1 if bad else 0)
