#!/usr/bin/env python3
"""Developer: cooperative tiles on / off (GPSAT_DEBUG_COOP) on the same batch: same bytes?  time?"""
import hashlib
import os
os.environ.setdefault("GPSAT_DEVELOPER", "1")     # GPSAT_DEBUG_* knobs are read in developer mode only
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402


def digest(r):
    return hashlib.sha256(r.theta.tobytes() + r.nll.tobytes() + np.asarray(r.f_mean).tobytes() + np.asarray(r.f_var).tobytes() +
                          r.n_eval.tobytes() + r.status.tobytes()).hexdigest()[:16]


def run(eng, b, kid, coop, **kw):
    os.environ["GPSAT_DEBUG_COOP"] = str(int(coop))
    T = b["T"]
    lo, hi = syn.default_bounds(T, b["D"])
    t0 = time.perf_counter()
    r = eng.fit_predict_batch(D=b["D"], obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                              theta0=np.ones((T, b["D"] + 2)), lo=lo, hi=hi, kernel=kid, **kw)
    return r, time.perf_counter() - t0


eng = Engine(0)
cases = [("1 x 1024, 3 iters", [1024], 2, dict(optimiser="lbfgs", max_iter=3)),
         ("1 x 2048, 5 iters", [2048], 0, dict(optimiser="lbfgs", max_iter=5)),
         ("1 x 2048 objective+gradient only", [2048], 2, dict(optimiser="none", want_grad=True)),
         ("ragged 24", [2048, 1536, 1024, 1024, 768, 640, 512, 512, 500, 400, 384, 300, 256, 200, 128, 100, 64, 33, 32, 31, 1, 700, 900, 1200], 2,
          dict(optimiser="lbfgs", max_iter=6))]
for n in (500, 400, 900, 1200, 1184, 640, 768, 1536, 416, 448):
    cases.append((f"single {n}", [n], 2, dict(optimiser="lbfgs", max_iter=4)))
which = sys.argv[1:] or [c[0] for c in cases]
for name, Ns, kid, kw in cases:
    if name not in which and not any(w in name for w in which):
        continue
    b = syn.make_batch(len(Ns), Ns, 40, 3, kid, base_seed=77)
    r0, t0 = run(eng, b, kid, False, **kw)
    r1, t1 = run(eng, b, kid, True, **kw)
    r2, t2 = run(eng, b, kid, True, **kw)
    same = digest(r0) == digest(r1) == digest(r2)
    print(f"{name}: off {r0.kernel_ms:.1f} ms, on {r1.kernel_ms:.1f} / {r2.kernel_ms:.1f} ms, evals {r0.n_eval.tolist()[:6]} status {np.unique(r1.status).tolist()}"
          f" bit-identical {same}", flush=True)
    if not same:
        print("   off", digest(r0), "on", digest(r1), digest(r2), "max |dtheta|", np.abs(r0.theta - r1.theta).max(), "nll", r0.nll[:3], r1.nll[:3])

# one evaluation, cooperative code path forced (GPSAT_DEBUG_COOP=2) against the plain one: which output differs?
for n in (416, 640, 500, 1200, 448):
    b = syn.make_batch(1, [n], 40, 3, 2, base_seed=77)
    kw = dict(optimiser="none", want_grad=True)
    r0, _ = run(eng, b, 2, 0, **kw)
    r2, _ = run(eng, b, 2, 2, **kw)
    os.environ["GPSAT_DEBUG_GRID"] = "1"
    r3, _ = run(eng, b, 2, 2, **kw)          # forced cooperative path, no helper workgroup in the launch at all
    del os.environ["GPSAT_DEBUG_GRID"]
    print(f"eval {n}: nll {r0.nll[0]!r} {r2.nll[0]!r} {r3.nll[0]!r} grad diff forced {np.abs(r0.grad - r2.grad).max():.3e} alone {np.abs(r0.grad - r3.grad).max():.3e}"
          f" fmean diff {np.abs(r0.f_mean - r2.f_mean).max():.3e} {np.abs(r0.f_mean - r3.f_mean).max():.3e}", flush=True)
    print("    grad", r0.grad[0], r2.grad[0], r3.grad[0])
