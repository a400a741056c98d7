"""GPU: teams of workgroups on large fp64 tiles (gpsat_kernels_f64.hip, "Teams"): with few large tiles, G workgroups run
every tile together, bulk-synchronously, through team barriers in device memory.  A team returns the bits one workgroup
returns ("helped vs not helped"), and the reference's own published large fit -- N = 2500, 1-D RBF,
likelihood variance fixed (docs/notebooks/using_gpus.ipynb:77,165: optimise 5.645 s on an RTX 2080) -- takes less than a
quarter of a second.  GPSAT_DEBUG_TEAM: workgroups per tile (1 = no teams; default: chosen by the library)."""
import os

import numpy as np
import pytest

from gpsat_amd import synthetic as syn

pytestmark = pytest.mark.gpu
os.environ["GPSAT_DEVELOPER"] = "1"       # the GPSAT_DEBUG_* knobs used below are read in developer mode only


@pytest.fixture(scope="module")
def eng():
    from gpsat_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
    os.environ.pop("GPSAT_DEBUG_TEAM", None)


def _run(eng, team, **kw):
    if team is None:
        os.environ.pop("GPSAT_DEBUG_TEAM", None)
    else:
        os.environ["GPSAT_DEBUG_TEAM"] = str(team)
    try:
        return eng.fit_predict_batch(dtype="f64", **kw)
    finally:
        os.environ.pop("GPSAT_DEBUG_TEAM", None)


def _same(a, b):
    for f in ("theta", "nll", "status", "n_eval", "n_iter", "f_mean", "f_var", "y_var"):
        np.testing.assert_array_equal(np.asarray(getattr(a, f)), np.asarray(getattr(b, f)), err_msg=f)
    if a.grad is not None:
        np.testing.assert_array_equal(a.grad, b.grad)
    if a.f_cov is not None:
        np.testing.assert_array_equal(np.asarray(a.f_cov), np.asarray(b.f_cov))


@pytest.mark.parametrize("kernel,Ns,extra", [
    ("RBF", [2000], dict(optimiser="lbfgs", max_iter=3, want_grad=True)),
    ("Matern32", [1500, 1100, 2047, 0, 1793], dict(optimiser="lbfgs", max_iter=2)),
    ("Matern52", [1601], dict(optimiser="none", want_grad=True, full_cov=True)),
    ("Matern12", [1280], dict(optimiser="adam", max_iter=3, adam_lr=0.05)),
])
def test_team_returns_the_bits_of_one_workgroup(eng, kernel, Ns, extra):
    kid = {"RBF": 0, "Matern12": 1, "Matern32": 2, "Matern52": 3}[kernel]
    b = syn.make_batch(len(Ns), Ns, 60, 3, kid, base_seed=11, dtype=np.float64)
    T = len(Ns)
    lo, hi = syn.default_bounds(T, 3)
    kw = dict(D=3, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=np.full((T, 5), 1.5),
              lo=lo, hi=hi, kernel=kernel, **extra)
    solo = _run(eng, 1, **kw)
    for g in (2, 5, 16, None):
        _same(solo, _run(eng, g, **kw))
    assert np.isin(solo.status[np.array(Ns) > 0], (0, 1, 5)).all()


def test_reference_published_n2500_fit_under_a_quarter_second(eng):
    np.random.seed(0)
    N, Lh, noise_std = 2500, 5, 0.05
    X = np.random.uniform(-Lh, Lh, (N, 1))
    y = np.cos(X[:, 0]) + noise_std * np.random.randn(N)
    y = y - y.mean()
    Xs = np.linspace(-Lh, Lh, 100)[:, None]
    kw = dict(D=1, obs_off=[0, N], X=X, y=y, pred_off=[0, 100], Xs=Xs, theta0=[[1.0, 1.0, noise_std ** 2]], trainable=[1, 1, 0],
              kernel="RBF", optimiser="lbfgs", max_iter=10_000)
    solo = _run(eng, 1, **kw)
    _run(eng, None, **kw)                                          # buffers of the team launch allocated
    team = _run(eng, None, **kw)
    _same(solo, team)
    assert team.status[0] == 0 and 8 <= team.n_eval[0] <= 40
    print(f"N=2500 fp64 fit + predict: one workgroup {solo.kernel_ms:.0f} ms, team {team.kernel_ms:.0f} ms (call {team.total_ms:.0f} ms), "
          f"{int(team.n_eval[0])} evaluations (reference on an RTX 2080: 5645 + 176 ms)")
    assert team.kernel_ms < 250.0 and team.total_ms < 250.0
    assert team.kernel_ms < 0.4 * solo.kernel_ms
