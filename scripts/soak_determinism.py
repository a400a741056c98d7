"""Developer: soak test of the dataflow sweep and the time-sliced queue -- random ragged batches run several times, with
and without slicing: every run of a batch on the same build must return the same bytes (the 4- and 8-wave builds sum the
gradient over different numbers of waves and may differ from each other in the last bits)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn

names = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}
bad = 0
for seed in range(int(os.environ.get("SEEDS", 4))):
    rng = np.random.default_rng(100 + seed)
    T, D, kid = int(rng.integers(600, 1500)), int(rng.integers(1, 4)), int(rng.integers(0, 4))
    big = rng.random() < 0.5
    Ns = rng.integers(10, 700 if not big else 1200, size=T).tolist()
    Ps = rng.integers(0, 40, size=T).tolist()
    b = syn.make_batch(T, Ns, Ps, D, kid, base_seed=1000 * seed)
    lo, hi = syn.default_bounds(T, D)
    kw = dict(D=D, kernel=names[kid], optimiser="lbfgs", max_iter=10, want_grad=True, obs_off=b["obs_off"], X=b["X"], y=b["y"],
              pred_off=b["pred_off"], Xs=b["Xs"], theta0=np.ones((T, D + 2)), lo=lo, hi=hi)
    for wg in (2, 1):
        ref = None
        eng = Engine(0, workgroups_per_cu=wg)
        for seg in ("0", "1", "4096", "0"):
            os.environ["GPSAT_DEBUG_SEG"] = seg
            r = eng.fit_predict_batch(**kw)
            sig = b"".join(getattr(r, n).tobytes() for n in ("theta", "nll", "grad", "status", "n_eval", "f_mean", "f_var", "y_var"))
            if ref is None:
                ref = sig
            same = sig == ref
            bad += not same
            print(f"seed {seed} T={T} D={D} {names[kid]} maxN={max(Ns)} wg/cu={wg} seg={seg}: {'same' if same else 'DIFFERENT'}; "
                  f"status counts {np.bincount(r.status, minlength=7).tolist()}", flush=True)
        del eng
print("mismatches:", bad)
sys.exit(1 if bad else 0)
