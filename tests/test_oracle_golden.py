"""Pin the CPU oracle (oracle/gp_oracle.py) against the reference's own known answers.

* tests/test_localexperts.py:203-227 (reference): lengthscale, LML, f*, f*_var to 1e-6
* GPSat/models/pure_python_gpr.py SGPkernel/SMLII_mod/GPR outputs (tests/golden/ref_purepython_*.npz)
* docs/notebooks/gp_regression.ipynb printed values
* tests/test_utils.py:962-1023 transform tolerances (1e-14 / 1e-12)
"""
import os

import numpy as np
import pytest

from oracle import gp_oracle as go


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_kat_sklearn_matern32(golden_dir):
    """Exactly the assertions of the reference's test_gpflow_gpr (tol 1e-6)."""
    g = _load(golden_dir, "kat_sklearn_matern32.npz")
    m = go.OracleGPR(coords=g["x_train"], obs=g["y_train"], obs_mean=None, kernel="Matern32")
    m.set_parameters(likelihood_variance=float(g["eps"]) ** 2)
    m.set_parameter_constraints({"lengthscales": {"low": 1e-10, "high": 5.0}})
    ok = m.optimise_parameters(fixed_params=["likelihood_variance", "kernel_variance"])
    out = m.predict(np.array([[float(g["x_test"])]]))
    tol = 1e-6
    assert ok
    assert abs(m.get_parameters()["lengthscales"][0] - float(g["ls"])) < tol
    assert abs(-m.get_objective_function_value() - float(g["ml"])) < tol
    assert abs(out["f*"][0] - float(g["pred_mean"])) < tol
    assert abs(out["f*_var"][0] - float(g["pred_std"]) ** 2) < tol
    assert abs(out["y_var"][0] - (float(g["pred_std"]) ** 2 + 1e-4)) < tol
    # full posterior covariance against sklearn's return_cov at the same optimum
    m.set_parameters(lengthscales=float(g["ls"]))
    oc = m.predict(g["cov_x"][:, None], full_cov=True)
    np.testing.assert_allclose(oc["f*_cov"], g["cov"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(oc["f*"], g["cov_mean"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.diag(oc["y_cov"]), oc["y_var"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(oc["y_cov"] - oc["f*_cov"], 1e-4 * np.eye(len(g["cov_x"])), rtol=0, atol=1e-15)


@pytest.mark.parametrize("N", [16, 128, 500])
def test_reference_numpy_functions(golden_dir, N):
    g = _load(golden_dir, f"ref_purepython_matern32_N{N}.npz")
    x, y, xs = g["x"], g["y"], g["xs"]
    K8 = go.kernel_matrix(2, x[:8], x[:8], np.array([3.0, 4.5, 5.0]), 0.7)
    np.testing.assert_allclose(K8, g["K8"], rtol=0, atol=1e-13)
    for i, th in enumerate(g["thetas"]):
        nll, grad = go.nll_and_grad(2, x, y, th)
        assert abs(nll - g["nll"][i]) < 1e-8 * max(1.0, abs(g["nll"][i]))
        # gradient oracle = central differences of the REFERENCE NLL
        np.testing.assert_allclose(grad, g["grads_fd"][i], rtol=2e-5, atol=2e-5)
        f, fv, yv = go.predict(2, x, y, xs, th)
        np.testing.assert_allclose(f, g["pred_mean"][i], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.sqrt(fv), g["pred_std"][i], rtol=0, atol=1e-8)


def test_notebook_rbf_kat(golden_dir):
    g = _load(golden_dir, "kat_notebook_rbf.npz")
    X, y = g["X"][:, None], g["y"]
    # printed kernel_variance is amplitude**2 of sklearn's ConstantKernel (sklearn_models.py:96)
    th0 = np.array([1.0, np.sqrt(1.5), 0.0025])
    nll0, _ = go.nll_and_grad(0, X, y, th0)
    assert abs(-nll0 - float(g["lml_init"])) < 5e-4
    f, fv, _ = go.predict(0, X, y, g["X_grid"][:, None], th0)
    assert abs(np.mean((g["f_truth"] - f) ** 2) - float(g["mse_init"])) < 1e-4
    # optimise lengthscale + kernel variance with likelihood variance fixed
    m = go.OracleGPR(coords=X, obs=y, kernel="RBF", kernel_kwargs={"variance": np.sqrt(1.5)},
                     noise_variance=0.0025)
    assert m.optimise_parameters(fixed_params=["likelihood_variance"])
    p = m.get_parameters()
    assert abs(p["lengthscales"][0] - float(g["ls_opt"])) < 2e-4
    assert abs(p["kernel_variance"] ** 2 - float(g["kv_printed_opt"])) < 2e-4
    assert abs(-m.get_objective_function_value() - float(g["lml_opt"])) < 5e-4


def test_transforms(golden_dir):
    g = _load(golden_dir, "transforms.npz")
    x = g["x_softplus"]
    y = go.softplus(x)
    np.testing.assert_array_almost_equal(y, g["softplus"], decimal=14)
    np.testing.assert_array_almost_equal(go.softplus(x, shift=10.0), y + 10.0, decimal=14)
    np.testing.assert_array_almost_equal(go.inverse_softplus(y), x, decimal=14)
    assert go.inverse_softplus(np.array(-1.0)) == -np.inf
    x2 = g["x_sigmoid"]
    s = go.sigmoid(x2)
    np.testing.assert_array_almost_equal(s, g["sigmoid"], decimal=14)
    np.testing.assert_array_almost_equal(go.inverse_sigmoid(s), x2, decimal=12)
    s2 = go.sigmoid(x2, -1, 1)
    np.testing.assert_array_almost_equal(s2, g["sigmoid"] * 2 - 1, decimal=14)
    np.testing.assert_array_almost_equal(go.inverse_sigmoid(s2, -1, 1), x2, decimal=12)
    assert go.inverse_sigmoid(np.array(-1.5), -1.0, 2.0) == -np.inf
    assert go.inverse_sigmoid(np.array(2.0), -1.0, 2.0) == np.inf


def test_gradient_chain_all_kernels():
    """dNLL/du by the analytic chain equals central differences for every kernel / transform."""
    rng = np.random.default_rng(5)
    N, D = 40, 3
    X = rng.uniform(-3, 3, (N, D))
    y = rng.standard_normal(N)
    lo = np.array([1e-8, 1e-8, np.nan, np.nan, np.nan])
    hi = np.array([12.0, 9.0, np.nan, np.nan, np.nan])
    shift = np.array([0, 0, 0, 0, go.LIK_VAR_LOWER])
    u = rng.standard_normal(D + 2) * 0.5
    for kid in range(4):
        def f(u_):
            th = go.theta_from_u(u_, lo, hi, shift)
            return go.nll_and_grad(kid, X, y, th, want_grad=False)[0]
        th = go.theta_from_u(u, lo, hi, shift)
        _, g = go.nll_and_grad(kid, X, y, th)
        gu = g * go.dtheta_du(th, lo, hi, shift)
        fd = np.array([(f(u + 1e-6 * e) - f(u - 1e-6 * e)) / 2e-6 for e in np.eye(D + 2)])
        np.testing.assert_allclose(gu, fd, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(go.u_from_theta(go.theta_from_u(u, lo, hi, shift), lo, hi, shift), u, atol=1e-9)


def test_transforms_against_reference_functions(golden_dir):
    """The oracle's transforms against the outputs of the reference's OWN functions (GPSat/utils.py:2320-2400, generated
    by tests/golden/make_golden.py), at the tolerances of the reference's tests (tests/test_utils.py:962-1023),
    including the out-of-range values (-inf at / below the lower bound, +inf at / above the upper one)."""
    g = _load(golden_dir, "transforms.npz")
    np.testing.assert_array_almost_equal(go.softplus(g["x_softplus"]), g["ref_softplus"], decimal=14)
    np.testing.assert_array_almost_equal(go.softplus(g["x_softplus"], shift=10.0), g["ref_softplus_shift10"], decimal=14)
    np.testing.assert_array_almost_equal(go.sigmoid(g["x_sigmoid"]), g["ref_sigmoid"], decimal=14)
    lo, hi = g["box"]
    np.testing.assert_array_almost_equal(go.sigmoid(g["x_sigmoid"], lo, hi), g["ref_sigmoid_box"], decimal=14)
    for y, ref, kw in ((g["y_inv_softplus"], g["ref_inv_softplus"], {}),
                       (g["y_inv_softplus_shift10"], g["ref_inv_softplus_shift10"], {"shift": 10.0})):
        got = go.inverse_softplus(y, **kw)
        fin = np.isfinite(ref)
        assert (np.isfinite(got) == fin).all() and (got[~fin] == ref[~fin]).all()
        np.testing.assert_allclose(got[fin], ref[fin], rtol=1e-14, atol=1e-14)
    got = go.inverse_sigmoid(g["y_inv_sigmoid_box"], lo, hi)
    ref = g["ref_inv_sigmoid_box"]
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all() and (got[~fin] == ref[~fin]).all()
    np.testing.assert_allclose(got[fin], ref[fin], rtol=0, atol=1e-12)


SK_KERNELS = [("Matern12", 1), ("Matern52", 3), ("RBF", 0)]


@pytest.mark.parametrize("name,kid", SK_KERNELS)
@pytest.mark.parametrize("N", [50, 500])
def test_sklearn_kernels_d3(golden_dir, name, kid, N):
    """Matern-1/2, Matern-5/2 and ARD RBF in D = 3 against scikit-learn -- the oracle of the reference's own test
    (tests/test_localexperts.py:40-49,203-227) -- at that test's tolerance (1e-6): objective, its gradient,
    predictive mean / variance / full covariance."""
    g = _load(golden_dir, "kat_sklearn_kernels.npz")
    X, y, Xs = g[f"X_{N}"], g[f"y_{N}"], g[f"Xs_{N}"]
    for i, th in enumerate(g[f"thetas_{N}"]):
        nll, grad = go.nll_and_grad(kid, X, y, th)
        assert abs(-nll - g[f"{name}_{N}_lml"][i]) < 1e-6 * max(1.0, abs(nll))
        # sklearn: d LML / d log(sf2, l1, l2, l3)  ->  dNLL/dtheta_j = -(.)/theta_j
        dl = g[f"{name}_{N}_dlml_dlog"][i]
        ref_grad = np.concatenate([-dl[1:4] / th[:3], [-dl[0] / th[3]]])
        np.testing.assert_allclose(grad[:4], ref_grad, rtol=1e-6, atol=1e-6)
        f, fv, yv = go.predict(kid, X, y, Xs, th)
        np.testing.assert_allclose(f, g[f"{name}_{N}_mean"][i], rtol=0, atol=1e-6)
        np.testing.assert_allclose(fv, g[f"{name}_{N}_std"][i] ** 2, rtol=0, atol=1e-6)
        fc, _ = go.predict_cov(kid, X, y, Xs, th)
        np.testing.assert_allclose(fc, g[f"{name}_{N}_cov"][i], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,kid", SK_KERNELS)
def test_sklearn_optimum_d3(golden_dir, name, kid):
    """The multi-parameter optimum sklearn's own L-BFGS-B finds (4 trainable scalars, noise fixed) is the oracle's: run to
    the gradient tolerance, length scales to 1e-3 and variances to 1e-6 as the reference's integration check asks
    (tests/integration.py:109-132)."""
    g = _load(golden_dir, "kat_sklearn_kernels.npz")
    X, y, Xs = g["X_50"], g["y_50"], g["Xs_50"]
    ref = g[f"{name}_50_opt_theta"]
    m = go.OracleGPR(X, y, kernel=name, noise_variance=0.01)
    m.set_parameter_constraints({"lengthscales": {"low": [1e-2] * 3, "high": [1e2] * 3}})
    assert m.optimise_parameters(fixed_params=["likelihood_variance"], tol=1e-14)
    assert abs(-m.get_objective_function_value() - float(g[f"{name}_50_opt_lml"])) < 1e-7
    th = m.theta
    np.testing.assert_allclose(th[:3], ref[:3], rtol=0, atol=1e-3)      # sklearn itself stops at SciPy's default ftol
    assert abs(th[3] - ref[3]) < 1e-5
    f, fv, _ = go.predict(kid, X, y, Xs, th)
    np.testing.assert_allclose(f, g[f"{name}_50_opt_mean"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(np.sqrt(fv), g[f"{name}_50_opt_std"], rtol=0, atol=1e-4)


def test_adam_matches_torch_optim_adam():
    """The oracle's Adam is torch.optim.Adam -- the optimiser of the reference's Adam path
    (GPSat/models/gpytorch_models.py:187-199: torch.optim.Adam(lr=0.1), one gradient per step) -- step for step, on the
    oracle's own objective and transforms (20 steps, D = 3 ARD, box-constrained length scales)."""
    import torch
    from gpsat_amd import synthetic as syn
    X, y, _, _ = syn.make_tile(77, 60, 0, 3, 0)
    lo = np.array([1e-8, 1e-8, 1e-8, -np.inf, -np.inf])
    hi = np.array([12.0, 12.0, 9.0, np.inf, np.inf])
    shift = np.array([0, 0, 0, 0, go.LIK_VAR_LOWER])
    u0 = go.u_from_theta(np.ones(5), lo, hi, shift)

    def fun(u):
        th = go.theta_from_u(u, lo, hi, shift)
        f, g = go.nll_and_grad(0, X, y, th)
        return f, g * go.dtheta_du(th, lo, hi, shift)

    u, ok, traj = go.adam_minimise(fun, u0, 20, lr=0.1)
    assert ok and traj.shape == (21, 5)
    p = torch.tensor(u0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam([p], lr=0.1)
    for k in range(20):
        opt.zero_grad()
        p.grad = torch.tensor(fun(p.detach().numpy())[1])
        opt.step()
        np.testing.assert_allclose(traj[k + 1], p.detach().numpy(), rtol=1e-12, atol=1e-13)
    assert fun(u)[0] < fun(u0)[0]


def test_adam_model_method_counts_evaluations():
    from gpsat_amd import synthetic as syn
    X, y, _, _ = syn.make_tile(5, 40, 0, 2, 2)
    m = go.OracleGPR(X, y, kernel="Matern32")
    f0 = m.get_objective_function_value()
    assert m.optimise_parameters_adam(max_iter=7, lr=0.05, fixed_params=["likelihood_variance"])
    assert m.n_eval == 8 and m.theta[3] == 1.0 and m.get_objective_function_value() < f0
