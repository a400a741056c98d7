#!/usr/bin/env python3
"""Developer: fp64 teams on / off (GPSAT_DEBUG_TEAM=1 = one workgroup per tile) on the same batch: same bytes?  time?"""
import hashlib
import os
os.environ.setdefault("GPSAT_DEVELOPER", "1")     # GPSAT_DEBUG_* knobs are read in developer mode only
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402


def digest(r):
    return hashlib.sha256(r.theta.tobytes() + r.nll.tobytes() + np.asarray(r.f_mean).tobytes() + np.asarray(r.f_var).tobytes() +
                          r.n_eval.tobytes() + r.status.tobytes() + (r.grad.tobytes() if r.grad is not None else b"")).hexdigest()[:16]


def run(eng, b, kid, team, **kw):
    if team is None:
        os.environ.pop("GPSAT_DEBUG_TEAM", None)
    else:
        os.environ["GPSAT_DEBUG_TEAM"] = str(team)
    T = b["T"]
    lo, hi = syn.default_bounds(T, b["D"])
    return eng.fit_predict_batch(D=b["D"], obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"],
                                 theta0=np.ones((T, b["D"] + 2)), lo=lo, hi=hi, kernel=kid, dtype="f64", **kw)


eng = Engine(0)
# one evaluation (objective + gradient) at fixed parameters: which output differs?
for n in (1600, 2000):
    b = syn.make_batch(1, [n], 50, 3, 0, base_seed=3, dtype=np.float64)
    r1 = run(eng, b, 0, 1, optimiser="none", want_grad=True)
    for g in (2, 8):
        r = run(eng, b, 0, g, optimiser="none", want_grad=True)
        print(f"eval N={n} team {g}: nll {r1.nll[0]!r} {r.nll[0]!r} grad diff {np.abs(r1.grad - r.grad).max():.3e} (|g| {np.abs(r1.grad).max():.3e})"
              f" f* diff {np.abs(r1.f_mean - r.f_mean).max():.3e} var diff {np.abs(r1.f_var - r.f_var).max():.3e}", flush=True)
        print("     ", r1.grad[0], r.grad[0])
cases = [("1 x 1024", [1024], 0, dict(optimiser="lbfgs", max_iter=3, want_grad=True)),
         ("1 x 2000", [2000], 2, dict(optimiser="lbfgs", max_iter=3, want_grad=True)),
         ("1 x 2500", [2500], 0, dict(optimiser="lbfgs", max_iter=20)),
         ("3 ragged", [1500, 1100, 2047], 3, dict(optimiser="lbfgs", max_iter=3)),
         ("1 x 2000 predict only", [2000], 0, dict(optimiser="none"))]
for name, Ns, kid, kw in cases:
    b = syn.make_batch(len(Ns), Ns, 100, 3, kid, base_seed=11, dtype=np.float64)
    r0 = run(eng, b, kid, 1, **kw)
    res = {1: r0}
    out = [f"solo {r0.kernel_ms:.1f} ms"]
    ok = True
    for g in (2, 4, 8, None):
        r = run(eng, b, kid, g, **kw)
        ok = ok and digest(r) == digest(r0)
        out.append(f"team {g if g else 'auto'}: {r.kernel_ms:.1f} ms" + ("" if digest(r) == digest(r0) else " DIFFERENT"))
    print(f"{name}: " + ", ".join(out) + f", evals {r0.n_eval.tolist()} status {r0.status.tolist()} bit-identical {ok}", flush=True)
