"""
CPU ORACLE (fp64, NumPy/SciPy) for the local-expert exact-GP hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``gpsat_amd``) never imports anything from ``oracle/``
and fails loudly when the HIP library is missing.

It restates, function by function, the algorithm the reference runs per tile:

* data intake / scaling ............ GPSat/models/base_model.py:134-245
* model defaults ................... GPSat/models/gpflow_models.py:116-157
* box constraints (sigmoid) ........ GPSat/models/gpflow_models.py:416-494,592-628
* optimise (L-BFGS-B on u) ......... GPSat/models/gpflow_models.py:291-329
* objective (NLL) .................. GPSat/models/gpflow_models.py:334-337,
                                     GPSat/models/pure_python_gpr.py:485-487
* analytic gradient structure ...... GPSat/models/pure_python_gpr.py:488-498
* predict dict ..................... GPSat/models/gpflow_models.py:187-273,
                                     GPSat/models/pure_python_gpr.py:540-549
* transforms ....................... GPSat/utils.py:2320-2400

The arithmetic of the default backend lives in third-party packages that are
NOT vendored in the reference (gpflow>=2.9.0, tensorflow>=2.14,<2.16,
tensorflow-probability<0.24, scipy L-BFGS-B: requirements.txt:12,15,29,34).
Their published algorithm is restated here and PINNED by
``tests/test_oracle_golden.py`` against

* the reference's own known-answer test tests/test_localexperts.py:22-49,203-227
  (sklearn Matern-3/2 fixture: lengthscale, LML, f*, f*_var),
* the reference's NumPy functions SGPkernel / SMLII_mod / GPR
  (GPSat/models/pure_python_gpr.py:378-553) evaluated in the build container
  (fixtures under tests/golden/, generator tests/golden/make_golden.py),
* printed notebook known-answers (docs/notebooks/gp_regression.ipynb:452,481-486).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_solve, solve_triangular
from scipy.optimize import minimize

KERNEL_IDS = {"RBF": 0, "SquaredExponential": 0, "Matern12": 1, "Exponential": 1,
              "Matern32": 2, "Matern52": 3}

LIK_VAR_LOWER = 1e-6          # GPflow Gaussian likelihood variance lower bound
_SQ_FLOOR = 1e-36             # GPflow's sqrt(max(r2, 1e-36)) for Matern kernels


# --------------------------------------------------------------------------
# transforms (GPSat/utils.py:2320-2400)
# --------------------------------------------------------------------------
def softplus(x, shift=0.0):
    # utils.py:2320-2323 (stable form)
    x = np.asarray(x, dtype=np.float64)
    return np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0) + shift


def inverse_softplus(y, shift=0.0):
    # utils.py:2326-2374: -inf at/below the shift, log(y) for tiny y, y for large y
    y = np.asarray(y, dtype=np.float64)
    y_ = y - shift
    thr = np.log(np.finfo(np.float64).eps) + 2.0
    out = np.empty_like(y_)
    with np.errstate(divide="ignore", invalid="ignore"):
        mid = np.log(-np.expm1(-y_)) + y_
        out = np.where(y_ <= 0, -np.inf,
                       np.where(y_ < np.exp(thr), np.log(np.where(y_ > 0, y_, 1.0)),
                                np.where(y_ > -thr, y_, mid)))
    return out


def sigmoid(x, low=0.0, high=1.0):
    # utils.py:2377-2380
    x = np.asarray(x, dtype=np.float64)
    return (high - low) / (1.0 + np.exp(-x)) + low


def inverse_sigmoid(y, low=0.0, high=1.0):
    # utils.py:2383-2400: -inf at/below low, +inf at/above high
    y = np.asarray(y, dtype=np.float64)
    low = np.broadcast_to(np.asarray(low, dtype=np.float64), y.shape)
    high = np.broadcast_to(np.asarray(high, dtype=np.float64), y.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        mid = -np.log((high - low) / (y - low) - 1.0)
    return np.where(y <= low, -np.inf, np.where(y >= high, np.inf, mid))


def theta_from_u(u, lo, hi, shift):
    """u (unconstrained) -> theta.  Box where lo/hi finite, else softplus + shift."""
    u = np.asarray(u, dtype=np.float64)
    box = np.isfinite(lo) & np.isfinite(hi)
    th_sp = softplus(u) + shift
    with np.errstate(over="ignore", invalid="ignore"):
        th_bx = np.where(box, lo, 0.0) + (np.where(box, hi, 1.0) - np.where(box, lo, 0.0)) / (1.0 + np.exp(-u))
    return np.where(box, th_bx, th_sp)


def u_from_theta(theta, lo, hi, shift):
    theta = np.asarray(theta, dtype=np.float64)
    box = np.isfinite(lo) & np.isfinite(hi)
    u_sp = inverse_softplus(theta, shift)
    u_bx = inverse_sigmoid(theta, np.where(box, lo, 0.0), np.where(box, hi, 1.0))
    return np.where(box, u_bx, u_sp)


def dtheta_du(theta, lo, hi, shift):
    """softplus -> 1 - exp(-(theta-shift)); box -> (th-lo)(hi-th)/(hi-lo)."""
    theta = np.asarray(theta, dtype=np.float64)
    box = np.isfinite(lo) & np.isfinite(hi)
    d_sp = -np.expm1(-(theta - shift))
    lo_ = np.where(box, lo, 0.0)
    hi_ = np.where(box, hi, 1.0)
    d_bx = (theta - lo_) * (hi_ - theta) / (hi_ - lo_)
    return np.where(box, d_bx, d_sp)


# --------------------------------------------------------------------------
# kernels  (Matern-3/2 cf. pure_python_gpr.py:393-394)
# --------------------------------------------------------------------------
def _scaled_sqdist(X, X2, ell):
    A = X / ell
    B = X2 / ell
    d = A[:, None, :] - B[None, :, :]
    return np.einsum("ijk,ijk->ij", d, d)


def kernel_matrix(kid, X, X2, ell, sf2):
    """k_theta(X, X2) without the noise term."""
    r2 = _scaled_sqdist(X, X2, ell)
    if kid == 0:
        return sf2 * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, _SQ_FLOOR))
    if kid == 1:
        return sf2 * np.exp(-r)
    if kid == 2:
        s = np.sqrt(3.0) * r
        return sf2 * (1.0 + s) * np.exp(-s)
    if kid == 3:
        s = np.sqrt(5.0) * r
        return sf2 * (1.0 + s + s * s / 3.0) * np.exp(-s)
    raise ValueError(f"kernel id {kid}")


def _g_over(kid, r2, sf2):
    """g(r) with dk/dl_d = g(r) * (x_d-x'_d)^2 / l_d^3   (SURVEY App. A)."""
    if kid == 0:
        return sf2 * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, _SQ_FLOOR))
    if kid == 1:
        with np.errstate(divide="ignore", invalid="ignore"):
            g = sf2 * np.exp(-r) / r
        return np.where(r2 > 0, g, 0.0)
    if kid == 2:
        return 3.0 * sf2 * np.exp(-np.sqrt(3.0) * r)
    if kid == 3:
        s = np.sqrt(5.0) * r
        return (5.0 * sf2 / 3.0) * (1.0 + s) * np.exp(-s)
    raise ValueError(kid)


# --------------------------------------------------------------------------
# objective, gradient, prediction
# --------------------------------------------------------------------------
def nll_and_grad(kid, X, y, theta, want_grad=True):
    """
    NLL = 1/2 y^T K^-1 y + sum log L_ii + N/2 log 2pi   (pure_python_gpr.py:485-487),
    K = k(X,X) + sn2 I.  Gradient w.r.t. theta = (l_1..l_D, sf2, sn2):
    1/2 sum_ab Q_ab dK_ab/dtheta_j with Q = K^-1 - alpha alpha^T
    (structure of pure_python_gpr.py:488-498, but w.r.t. the raw parameters).
    Returns (nll, grad) ; (inf, nan) when K is not positive definite.
    """
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    N, D = X.shape
    ell, sf2, sn2 = np.asarray(theta[:D], dtype=np.float64), float(theta[D]), float(theta[D + 1])
    Kf = kernel_matrix(kid, X, X, ell, sf2)
    K = Kf + sn2 * np.eye(N)
    try:
        L = np.linalg.cholesky(K)
    except np.linalg.LinAlgError:
        return np.inf, np.full(D + 2, np.nan)
    z = solve_triangular(L, y, lower=True)
    nll = 0.5 * z @ z + np.log(np.diag(L)).sum() + 0.5 * N * np.log(2 * np.pi)
    if not want_grad:
        return nll, None
    alpha = solve_triangular(L, z, lower=True, trans="T")
    Kinv = cho_solve((L, True), np.eye(N))
    Q = Kinv - np.outer(alpha, alpha)
    g = np.empty(D + 2)
    r2 = _scaled_sqdist(X, X, ell)
    G = _g_over(kid, r2, sf2)
    QG = Q * G
    for d in range(D):
        dd = (X[:, d][:, None] - X[:, d][None, :]) ** 2
        g[d] = 0.5 * np.sum(QG * dd) / ell[d] ** 3
    g[D] = 0.5 * np.sum(Q * Kf) / sf2
    g[D + 1] = 0.5 * np.trace(Q)
    return nll, g


def predict(kid, X, y, Xs, theta):
    """f* = K*^T alpha, f*_var = sf2 - colsum((L^-1 K*)^2), y_var = f*_var + sn2
    (gpflow_models.py:229-243; Alg. 2.1 R&W as pure_python_gpr.py:540-549)."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    Xs = np.asarray(Xs, dtype=np.float64)
    N, D = X.shape
    ell, sf2, sn2 = np.asarray(theta[:D], dtype=np.float64), float(theta[D]), float(theta[D + 1])
    K = kernel_matrix(kid, X, X, ell, sf2) + sn2 * np.eye(N)
    L = np.linalg.cholesky(K)
    z = solve_triangular(L, y, lower=True)
    Ks = kernel_matrix(kid, X, Xs, ell, sf2)
    V = solve_triangular(L, Ks, lower=True)
    f = V.T @ z
    fvar = sf2 - np.sum(V * V, axis=0)
    return f, fvar, fvar + sn2


def predict_cov(kid, X, y, Xs, theta):
    """Full posterior covariance f*_cov = K** - V^T V (GPflow predict_f(full_cov=True), as consumed by
    gpflow_models.py:245-263) and the predictive covariance y_cov = f*_cov + diag(y_var - f*_var)."""
    X = np.asarray(X, dtype=np.float64)
    Xs = np.asarray(Xs, dtype=np.float64)
    N, D = X.shape
    ell, sf2, sn2 = np.asarray(theta[:D], dtype=np.float64), float(theta[D]), float(theta[D + 1])
    L = np.linalg.cholesky(kernel_matrix(kid, X, X, ell, sf2) + sn2 * np.eye(N))
    V = solve_triangular(L, kernel_matrix(kid, X, Xs, ell, sf2), lower=True)
    f_cov = kernel_matrix(kid, Xs, Xs, ell, sf2) - V.T @ V
    f_var = np.diag(f_cov)
    y_cov = f_cov.copy()
    y_cov[np.arange(len(Xs)), np.arange(len(Xs))] += (f_var + sn2) - f_var
    return f_cov, y_cov


# --------------------------------------------------------------------------
# Adam (Kingma & Ba 2015, Algorithm 1) as torch.optim.Adam computes it -- the optimiser of the reference's Adam path
# (GPSat/models/gpytorch_models.py:187-199: torch.optim.Adam(lr=0.1), defaults betas (0.9, 0.999), eps 1e-8, one
# gradient evaluation per step).  Pinned against torch.optim.Adam itself in tests/test_oracle_golden.py.
# --------------------------------------------------------------------------
def adam_minimise(fun, u0, steps, lr=0.1, b1=0.9, b2=0.999, eps=1e-8):
    """fun(u) -> (f, g).  ``steps`` updates: u_k = u_{k-1} - lr * m_hat / (sqrt(v_hat) + eps) with the gradient at
    u_{k-1}; steps + 1 evaluations in all (the last one at the returned point).  A non-finite objective ends the run at
    the previous iterate.  Returns (u, ok, trajectory [steps + 1, len(u)])."""
    u = np.array(u0, dtype=np.float64)
    m = np.zeros_like(u)
    v = np.zeros_like(u)
    traj = [u.copy()]
    f, g = fun(u)
    if not np.isfinite(f):
        return u, False, np.array(traj)
    for k in range(1, steps + 1):
        m = b1 * m + (1.0 - b1) * g
        v = b2 * v + (1.0 - b2) * g * g
        mh = m / (1.0 - b1 ** k)
        vh = v / (1.0 - b2 ** k)
        un = u - lr * mh / (np.sqrt(vh) + eps)
        f, gn = fun(un)
        if not np.isfinite(f):
            return u, False, np.array(traj)
        u, g = un, gn
        traj.append(u.copy())
    return u, True, np.array(traj)


# --------------------------------------------------------------------------
# the per-tile model: defaults, constraints, optimise, predict
# --------------------------------------------------------------------------
class OracleGPR:
    """fp64 restatement of GPflowGPRModel's behaviour for one tile.

    Parameter vector order everywhere: theta = (l_1..l_D, kernel_variance, likelihood_variance).
    """

    param_names = ["lengthscales", "kernel_variance", "likelihood_variance"]

    def __init__(self, coords, obs, coords_scale=None, obs_scale=None, obs_mean=None,
                 kernel="Matern32", kernel_kwargs=None, noise_variance=None):
        coords = np.array(coords, dtype=np.float64)
        obs = np.array(obs, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords[:, None]
        obs = obs.reshape(len(coords), -1)
        assert not np.isnan(coords).any() and not np.isnan(obs).any()   # base_model.py:188-189
        # base_model.py:195-200: "local" -> column mean, anything else -> 0
        self.obs_mean = np.mean(obs, axis=0)[None, :] if isinstance(obs_mean, str) and obs_mean == "local" \
            else np.array([[0.0]])
        self.obs_scale = np.atleast_2d(1.0 if obs_scale is None else obs_scale).astype(np.float64)
        self.coords_scale = np.atleast_2d(1.0 if coords_scale is None else coords_scale).astype(np.float64)
        self.coords = coords / self.coords_scale                     # base_model.py:243
        self.obs = (obs - self.obs_mean) / self.obs_scale            # base_model.py:244-245
        self.kid = KERNEL_IDS[kernel]
        D = self.coords.shape[1]
        kk = dict(kernel_kwargs or {})
        ls = np.broadcast_to(np.asarray(kk.get("lengthscales", np.ones(D)), dtype=np.float64), (D,)).copy()
        self.theta = np.concatenate([ls, [float(kk.get("variance", 1.0))],
                                     [1.0 if noise_variance is None else float(noise_variance)]])
        self.lo = np.full(D + 2, -np.inf)
        self.hi = np.full(D + 2, np.inf)
        self.shift = np.zeros(D + 2)
        self.shift[D + 1] = LIK_VAR_LOWER
        self.D = D

    # -- accessors (gpflow_models.py:339-411)
    def get_parameters(self):
        D = self.D
        return {"lengthscales": self.theta[:D].copy(), "kernel_variance": float(self.theta[D]),
                "likelihood_variance": float(self.theta[D + 1])}

    def set_parameters(self, **kw):
        D = self.D
        if "lengthscales" in kw:
            self.theta[:D] = np.broadcast_to(np.asarray(kw["lengthscales"], dtype=np.float64), (D,))
        if "kernel_variance" in kw:
            self.theta[D] = float(np.asarray(kw["kernel_variance"]).reshape(-1)[0])
        if "likelihood_variance" in kw:
            v = float(np.asarray(kw["likelihood_variance"]).reshape(-1)[0])
            if (not np.isfinite(self.lo[D + 1])) and v < LIK_VAR_LOWER:
                v = LIK_VAR_LOWER                                     # gpflow_models.py:404-409
            self.theta[D + 1] = v

    def _slice(self, name):
        D = self.D
        return {"lengthscales": slice(0, D), "kernel_variance": slice(D, D + 1),
                "likelihood_variance": slice(D + 1, D + 2)}[name]

    def set_parameter_constraints(self, constraints, move_within_tol=True, tol=1e-8):
        # gpflow_models.py:416-494
        for name, c in constraints.items():
            sl = self._slice(name)
            n = sl.stop - sl.start
            low = np.atleast_1d(np.asarray(c["low"], dtype=np.float64))
            high = np.atleast_1d(np.asarray(c["high"], dtype=np.float64))
            low = np.broadcast_to(low, (n,)).copy() if len(low) == 1 and n == 1 else low
            high = np.broadcast_to(high, (n,)).copy() if len(high) == 1 and n == 1 else high
            assert len(low) == n and len(high) == n
            assert np.all(low <= high)
            if c.get("scale", False):
                sm = c.get("scale_magnitude", None)
                if sm is None:
                    low = low / self.coords_scale[0, :]
                    high = high / self.coords_scale[0, :]
                else:
                    low = low / sm
                    high = high / sm
            vals = self.theta[sl].copy()
            mwt = c.get("move_within_tol", move_within_tol)
            t = c.get("tol", tol)
            if mwt:
                half = np.min(high - low) / 2
                if t > half:
                    t = half
                vals[vals > (high - t)] = high[vals > (high - t)] - t
                vals[vals < (low + t)] = low[vals < (low + t)] + t
            self.theta[sl] = vals
            self.lo[sl] = low
            self.hi[sl] = high
            self.shift[sl] = 0.0

    # -- objective / optimise
    def get_objective_function_value(self):
        return nll_and_grad(self.kid, self.coords, self.obs[:, 0], self.theta, want_grad=False)[0]

    def optimise_parameters(self, max_iter=10_000, fixed_params=None, **opt_kwargs):
        """scipy L-BFGS-B over unconstrained u of the trainable entries
        (gpflow_models.py:291-329; GPflow Scipy optimiser => jac=True, options maxiter)."""
        D = self.D
        trainable = np.ones(D + 2, dtype=bool)
        for p in (fixed_params or []):
            trainable[self._slice(p)] = False
        self.n_eval = 0
        if not trainable.any():
            return True
        u_all = u_from_theta(self.theta, self.lo, self.hi, self.shift)

        def fun(u_tr):
            u = u_all.copy()
            u[trainable] = u_tr
            th = theta_from_u(u, self.lo, self.hi, self.shift)
            th[~trainable] = self.theta[~trainable]
            f, g = nll_and_grad(self.kid, self.coords, self.obs[:, 0], th)
            self.n_eval += 1
            if not np.isfinite(f):
                return 1e300, np.zeros(trainable.sum())
            gu = g * dtheta_du(th, self.lo, self.hi, self.shift)
            return f, gu[trainable]

        res = minimize(fun, u_all[trainable], jac=True, method="L-BFGS-B",
                       options=dict(maxiter=max_iter), **opt_kwargs)
        u = u_all.copy()
        u[trainable] = res.x
        th = theta_from_u(u, self.lo, self.hi, self.shift)
        th[~trainable] = self.theta[~trainable]
        self.theta = th
        self.opt_result = res
        return bool(res.success)

    def optimise_parameters_adam(self, max_iter=20, lr=0.1, fixed_params=None):
        """Exactly ``max_iter`` Adam steps on the unconstrained u of the trainable entries (``adam_minimise`` below).
        The reference's Adam path is ``torch.optim.Adam(lr=0.1)`` over the model parameters
        (GPSat/models/gpytorch_models.py:187-199); BASELINE.json's north_star names "20 L-BFGS/Adam steps".  Here it runs on
        this model's objective and transforms.  Returns False when an evaluation met a non-positive-definite K (the
        iterate before it is kept), else True."""
        D = self.D
        trainable = np.ones(D + 2, dtype=bool)
        for p in (fixed_params or []):
            trainable[self._slice(p)] = False
        self.n_eval = 0
        u_all = u_from_theta(self.theta, self.lo, self.hi, self.shift)

        def fun(u_tr):
            u = u_all.copy()
            u[trainable] = u_tr
            th = theta_from_u(u, self.lo, self.hi, self.shift)
            th[~trainable] = self.theta[~trainable]
            f, g = nll_and_grad(self.kid, self.coords, self.obs[:, 0], th)
            self.n_eval += 1
            if not np.isfinite(f):
                return np.inf, np.zeros(trainable.sum())
            return f, (g * dtheta_du(th, self.lo, self.hi, self.shift))[trainable]

        u_tr, ok, traj = adam_minimise(fun, u_all[trainable], max_iter, lr)
        u = u_all.copy()
        u[trainable] = u_tr
        th = theta_from_u(u, self.lo, self.hi, self.shift)
        th[~trainable] = self.theta[~trainable]
        self.theta = th
        self.adam_trajectory = traj
        return ok

    def predict(self, coords, full_cov=False, apply_scale=True):
        coords = np.asarray(coords, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords[None, :]
        if apply_scale:
            coords = coords / self.coords_scale
        f, fv, yv = predict(self.kid, self.coords, self.obs[:, 0], coords, self.theta)
        out = {"f*": f, "f*_var": fv, "y_var": yv, "f_bar": np.repeat(self.obs_mean[:, 0], len(f))}
        if full_cov:                                                  # gpflow_models.py:245-263
            out["f*_cov"], out["y_cov"] = predict_cov(self.kid, self.coords, self.obs[:, 0], coords, self.theta)
            out["f*_var"] = np.diag(out["f*_cov"]).copy()
        return out


# --------------------------------------------------------------------------
# batch helper used by tests / cpu_baseline (ragged CSR layout identical to the C ABI)
# --------------------------------------------------------------------------
def fit_predict_batch(kid, D, obs_off, X, y, pred_off, Xs, theta0, lo, hi, trainable,
                      max_iter=20, optimise=True):
    """Run the oracle on a packed ragged batch.  X, Xs already scaled; y de-meaned."""
    T = len(obs_off) - 1
    H = D + 2
    theta = np.zeros((T, H))
    nll = np.zeros(T)
    n_eval = np.zeros(T, dtype=np.int64)
    success = np.zeros(T, dtype=bool)
    fm = np.zeros(pred_off[-1])
    fv = np.zeros(pred_off[-1])
    yv = np.zeros(pred_off[-1])
    names = {0: "RBF", 1: "Matern12", 2: "Matern32", 3: "Matern52"}
    for t in range(T):
        a, b = obs_off[t], obs_off[t + 1]
        m = OracleGPR(X[a:b], y[a:b], kernel=names[kid])
        m.theta = np.array(theta0[t], dtype=np.float64)
        box = np.isfinite(lo[t]) & np.isfinite(hi[t])
        m.lo = np.where(box, lo[t], -np.inf)
        m.hi = np.where(box, hi[t], np.inf)
        m.shift = np.where(box, 0.0, m.shift)
        if optimise:
            fixed_mask = ~np.asarray(trainable, dtype=bool)
            # emulate fixed_params via a mask
            names_fixed = []
            if fixed_mask[:D].all():
                names_fixed.append("lengthscales")
            if fixed_mask[D]:
                names_fixed.append("kernel_variance")
            if fixed_mask[D + 1]:
                names_fixed.append("likelihood_variance")
            success[t] = m.optimise_parameters(max_iter=max_iter, fixed_params=names_fixed)
            n_eval[t] = m.n_eval
        theta[t] = m.theta
        nll[t] = m.get_objective_function_value()
        pa, pb = pred_off[t], pred_off[t + 1]
        if pb > pa:
            out = m.predict(Xs[pa:pb], apply_scale=False)
            fm[pa:pb], fv[pa:pb], yv[pa:pb] = out["f*"], out["f*_var"], out["y_var"]
    return dict(theta=theta, nll=nll, n_eval=n_eval, success=success, f_mean=fm, f_var=fv, y_var=yv)
