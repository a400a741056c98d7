"""GPU: the on-device Adam (gpsat_opt.h, GPSAT_OPT_ADAM) against the oracle's Adam (oracle/gp_oracle.py::adam_minimise,
itself pinned step for step against torch.optim.Adam -- the optimiser of the reference's Adam path,
GPSat/models/gpytorch_models.py:187-199 -- by tests/test_oracle_golden.py).  BASELINE.json's north_star names
"20 L-BFGS/Adam steps".

fp64 kernels: the end point of 20 steps to 1e-8 relative.  fp32 kernels: every step uses a gradient with the stated
fp64 -> fp32 error (2e-3 relative, tests/test_gpu_parity.py); Adam normalises the step by the gradient's running
magnitude, so 20 steps of size <= lr drift apart by at most ~ 20 * lr * 2e-3 in u -- asserted as 2e-2 absolute in u and
through the objective at the end point (1e-4 * N)."""
import numpy as np
import pytest

from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu

NAMES = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}


@pytest.fixture(scope="module")
def eng():
    from gpsat_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _oracle_adam(b, t, kid, th0, lo, hi, steps, lr, trainable):
    a, e = b["obs_off"][t], b["obs_off"][t + 1]
    m = go.OracleGPR(b["X"][a:e].astype(np.float64), b["y"][a:e].astype(np.float64), kernel=NAMES[kid])
    m.theta = np.array(th0, dtype=np.float64)
    box = np.isfinite(lo) & np.isfinite(hi)
    m.lo, m.hi = np.where(box, lo, -np.inf), np.where(box, hi, np.inf)
    m.shift = np.where(box, 0.0, m.shift)
    D = m.D
    fixed = []
    if not trainable[:D].all():
        fixed.append("lengthscales")
    if not trainable[D]:
        fixed.append("kernel_variance")
    if not trainable[D + 1]:
        fixed.append("likelihood_variance")
    ok = m.optimise_parameters_adam(max_iter=steps, lr=lr, fixed_params=fixed)
    return m, ok


@pytest.mark.parametrize("kid,D,N", [(0, 3, 200), (2, 3, 120), (3, 2, 90), (1, 1, 64)])
def test_fp64_adam_end_point_matches_oracle(eng, kid, D, N):
    T, P, steps, lr = 8 if N <= 120 else 4, 6, 20, 0.1
    b = syn.make_batch(T, N, P, D, kid, base_seed=300 + kid, dtype=np.float64)
    th0 = np.ones((T, D + 2))
    lo, hi = syn.default_bounds(T, D)
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=th0,
                              lo=lo, hi=hi, kernel=NAMES[kid], optimiser="adam", max_iter=steps, adam_lr=lr, dtype="f64")
    assert (r.status == 1).all() and (r.n_eval == steps + 1).all() and (r.n_iter == steps).all()
    for t in range(T):
        m, ok = _oracle_adam(b, t, kid, th0[t], lo[t], hi[t], steps, lr, np.ones(D + 2, bool))
        assert ok
        np.testing.assert_allclose(r.theta[t], m.theta, rtol=1e-8, atol=1e-12)
        assert abs(r.nll[t] - m.get_objective_function_value()) <= 1e-9 * max(1.0, abs(r.nll[t]))


def test_fp64_adam_with_fixed_parameter_and_other_rate(eng):
    T, N, P, D, kid, steps, lr = 4, 80, 3, 2, 2, 13, 0.03
    b = syn.make_batch(T, N, P, D, kid, base_seed=911, dtype=np.float64)
    th0 = np.tile([1.5, 0.7, 0.8, 0.2], (T, 1))
    nan = np.full((T, D + 2), np.nan)
    tr = np.array([1, 1, 1, 0], bool)
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=th0,
                              trainable=tr, kernel=NAMES[kid], optimiser="adam", max_iter=steps, adam_lr=lr, dtype="f64")
    assert (r.n_eval == steps + 1).all()
    for t in range(T):
        m, ok = _oracle_adam(b, t, kid, th0[t], nan[t], nan[t], steps, lr, tr)
        assert ok and m.theta[D + 1] == 0.2 and r.theta[t, D + 1] == 0.2
        np.testing.assert_allclose(r.theta[t], m.theta, rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("kid,D,N", [(0, 3, 500), (2, 3, 200)])
def test_fp32_adam_within_gradient_tolerance(eng, kid, D, N):
    T, P, steps, lr = 8, 5, 20, 0.1
    b = syn.make_batch(T, N, P, D, kid, base_seed=40 + kid)
    th0 = np.ones((T, D + 2))
    lo, hi = syn.default_bounds(T, D)
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=b["X"], y=b["y"], pred_off=b["pred_off"], Xs=b["Xs"], theta0=th0,
                              lo=lo, hi=hi, kernel=NAMES[kid], optimiser="adam", max_iter=steps, adam_lr=lr)
    assert (r.status == 1).all() and (r.n_eval == steps + 1).all()
    for t in range(T):
        m, ok = _oracle_adam(b, t, kid, th0[t], lo[t], hi[t], steps, lr, np.ones(D + 2, bool))
        assert ok
        u_gpu = go.u_from_theta(r.theta[t], m.lo, m.hi, m.shift)
        u_ref = go.u_from_theta(m.theta, m.lo, m.hi, m.shift)
        assert np.max(np.abs(u_gpu - u_ref)) <= 2e-2, (t, u_gpu, u_ref)
        # the objective at the GPU's end point, evaluated by the oracle, against the oracle's own end point
        a, e = b["obs_off"][t], b["obs_off"][t + 1]
        f_gpu = go.nll_and_grad(kid, b["X"][a:e].astype(np.float64), b["y"][a:e].astype(np.float64), r.theta[t], want_grad=False)[0]
        assert abs(f_gpu - m.get_objective_function_value()) <= 1e-4 * N
