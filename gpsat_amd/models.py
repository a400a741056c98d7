"""HipGPRModel -- per-tile model class with the reference's BaseGPRModel / GPflowGPRModel interface.

Drop-in hook: the reference selects its backend with ``model_config["oi_model"]``, either a
registry name or ``{"path_to_model": "gpsat_amd.models", "model_name": "HipGPRModel"}``
(GPSat/local_experts.py:319-325).  Every method the orchestrator calls on a model
(GPSat/local_experts.py:1043-1180) exists here with the same name, argument meaning and error
behaviour as GPSat/models/gpflow_models.py:26-663, but the arithmetic runs in the gfx950 HIP
kernels through the C ABI (include/gpsat_hip.h).  There is no CPU fallback.

Differences that are deliberate and documented in DESIGN.md:
  * compute dtype is fp32 on the GPU by default, ``dtype="f64"`` selects the fp64 kernels (host-side scaling /
    constraints are always fp64 like the reference);
  * ``mean_function`` and custom likelihoods are not built (NotImplementedError);
  * no TensorFlow import, no per-construction device probe (the engine knows its device).
"""
from __future__ import annotations

import platform
import re
import warnings
from typing import Dict, List

import numpy as np

try:  # pandas is optional at import time (arrays may be passed directly)
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None

from . import _lib as L

LIKELIHOOD_VARIANCE_LOWER_BOUND = 1e-6   # GPflow Gaussian likelihood default (gpflow_models.py:404-409)

_cpu_name_cache = None


def _processor_name():
    # base_model.py:302-323 (Linux branch), cached: the reference shells out per construction
    global _cpu_name_cache
    if _cpu_name_cache is None:
        name = platform.processor() or "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if "model name" in line:
                        name = re.sub(".*model name.*:", "", line, 1).strip()
                        break
        except OSError:
            pass
        _cpu_name_cache = name
    return _cpu_name_cache


def clamp_within(vals, lo, hi, tol):
    """Pull values to at least ``tol`` inside the box [lo, hi] (the effect of gpflow_models.py:471-479): ``tol`` is
    capped at half the narrowest width; the upper side is applied first.  ``vals`` may carry leading batch axes."""
    margin = min(float(tol), float(np.min(hi - lo)) / 2)
    vals = np.where(vals > hi - margin, hi - margin, vals)
    return np.where(vals < lo + margin, lo + margin, vals)


def _as_name_list(c):
    return [c] if isinstance(c, str) else c


def _as_columns(a):
    a = np.array(a)
    return a[:, None] if a.ndim == 1 else a


def _as_scale_row(s):
    """None -> [[1]]; number -> [[s]]; list -> one row; arrays pass through (base_model.py:212-232)."""
    if s is None:
        return np.ones((1, 1))
    if isinstance(s, (int, float, list)):
        return np.atleast_2d(np.asarray(s, dtype=np.float64))
    return np.asarray(s, dtype=np.float64)


class HipGPRModel:
    """Exact GP regression for one expert tile on MI355X (mirror of GPflowGPRModel)."""

    def __init__(self, data=None, coords_col=None, obs_col=None, coords=None, obs=None,
                 coords_scale=None, obs_scale=None, obs_mean=None, verbose=True, *,
                 kernel="Matern32", kernel_kwargs=None, mean_function=None, mean_func_kwargs=None,
                 noise_variance=None, likelihood=None, engine=None, dtype="f32", **kwargs):
        # ---- data intake (behaviour of GPSat/models/base_model.py:134-189): a frame + column names, or bare arrays
        if data is not None:
            if coords_col is None or obs_col is None:
                missing = "coord_col" if coords_col is None else "obs_col"
                raise AssertionError(f"data was provided, but {missing} was not")
            coords_col, obs_col = _as_name_list(coords_col), _as_name_list(obs_col)
            raw_coords, raw_obs = data.loc[:, coords_col].to_numpy(), data.loc[:, obs_col].to_numpy()
        else:
            for label, arr in (("obs", obs), ("coords", coords)):
                if arr is None:
                    raise AssertionError(f"data is {data}, and so is {label}: {arr}, provide either")
                if not isinstance(arr, np.ndarray):
                    raise AssertionError(f"if {label} is provided directly it must be an np.array")
            raw_obs, raw_coords = _as_columns(obs), _as_columns(coords)
            if len(raw_obs) != len(raw_coords):
                raise AssertionError("obs and coords lengths don't match ")
            coords_col = list(range(raw_coords.shape[1])) if coords_col is None else coords_col
            obs_col = [0] if obs_col is None else obs_col
        self.coords_col, self.obs_col = coords_col, obs_col
        for label, arr in (("coords", raw_coords), ("obs", raw_obs)):
            if np.isnan(arr).any():
                raise AssertionError(f"nans found in {label}")
        if raw_obs.shape[1] != 1:
            raise AssertionError("HipGPRModel handles a single observation column")

        # ---- de-mean / scale (base_model.py:195-245): only the string "local" selects the column mean, every other
        #      obs_mean (numbers and lists included) means zero; scales become (1, k) rows
        self.obs_mean = raw_obs.mean(axis=0, keepdims=True).astype(np.float64) \
            if (isinstance(obs_mean, str) and obs_mean == "local") else np.zeros((1, 1))
        self.obs_scale = _as_scale_row(obs_scale)
        self.coords_scale = _as_scale_row(coords_scale)
        self.coords = np.array(raw_coords, dtype=np.float64) / self.coords_scale
        self.obs = (np.array(raw_obs, dtype=np.float64) - self.obs_mean) / self.obs_scale

        # ---- kernel / defaults: gpflow_models.py:113-157
        assert kernel is not None, "kernel was not provided"
        if not isinstance(kernel, str) or kernel not in L.KERNEL_IDS:
            raise NotImplementedError(f"kernel {kernel!r}: this backend builds {sorted(L.KERNEL_IDS)}")
        if mean_function is not None or likelihood is not None:
            raise NotImplementedError("mean_function / custom likelihood are not built in the HIP backend")
        self.kernel = kernel
        if dtype not in ("f32", "f64"):
            raise ValueError("dtype must be 'f32' or 'f64'")
        self.dtype = dtype                                # device compute precision (the reference computes in fp64)
        D = self.coords.shape[1]
        if D > 4:
            raise NotImplementedError("HIP backend is built for 1..4 input dimensions")
        self.D = D
        kk = dict(kernel_kwargs or {})
        ls = np.broadcast_to(np.asarray(kk.get("lengthscales", np.ones(D)), dtype=np.float64), (D,)).copy()
        self._theta = np.concatenate([ls, [float(kk.get("variance", 1.0))],
                                      [1.0 if noise_variance is None else float(noise_variance)]])
        self._lo = np.full(D + 2, np.nan)
        self._hi = np.full(D + 2, np.nan)
        self._trainable = np.ones(D + 2, dtype=bool)

        # ---- device info: base_model.py:259 (attributes the orchestrator reads at local_experts.py:1180)
        from .engine import default_engine
        self._engine = engine if engine is not None else default_engine()
        self.gpu_name = self._engine.device_name
        self.cpu_name = _processor_name()
        self.n_eval = 0
        self.status = None

        # base_model.py:270-277
        for pn in self.param_names:
            assert not bool(re.search(" ", pn)), f"param_name: '{pn}' has a space (' ') in it, which is prohibited"
            getattr(self, f"set_{pn}")
            getattr(self, f"get_{pn}")

    # ------------------------------------------------------------------ interface
    @property
    def param_names(self) -> List[str]:
        return ["lengthscales", "kernel_variance", "likelihood_variance"]

    def get_parameters(self, *args, return_dict=True):
        # base_model.py:370-403
        if len(args) == 0:
            args = self.param_names
        for a in args:
            assert a in self.param_names, f"cannot get parameters for: {a}, it's not in param_names: {self.param_names}"
        if return_dict:
            return {a: getattr(self, f"get_{a}")() for a in args}
        return [getattr(self, f"get_{a}")() for a in args]

    def set_parameters(self, **kwargs):
        # base_model.py:405-422
        for k, v in kwargs.items():
            assert k in self.param_names, f"cannot get parameters for: {k}, it's not in param_names: {self.param_names}"
            getattr(self, f"set_{k}")(v)

    def set_parameter_constraints(self, constraints_dict, **kwargs):
        # base_model.py:424-439
        for k, v in constraints_dict.items():
            assert k in self.param_names, f"cannot get parameters for: {k}, it's not in param_names: {self.param_names}"
            getattr(self, f"set_{k}_constraints")(**v, **kwargs)

    # -- getters / setters: gpflow_models.py:339-411
    def get_lengthscales(self) -> np.ndarray:
        return self._theta[:self.D].copy()

    def get_kernel_variance(self) -> float:
        return float(self._theta[self.D])

    def get_likelihood_variance(self) -> float:
        return float(self._theta[self.D + 1])

    def set_lengthscales(self, lengthscales):
        v = np.asarray(lengthscales, dtype=np.float64).reshape(-1)
        assert len(v) in (1, self.D), f"lengthscales must have length 1 or {self.D}"
        self._theta[:self.D] = v

    def set_kernel_variance(self, kernel_variance):
        if isinstance(kernel_variance, np.ndarray):
            assert (len(kernel_variance) == 1) & (len(kernel_variance.shape) == 1), \
                f"set_kernel_variance expected to receive float, or np.array with len(1), shape:(1,), got" \
                f"len: {len(kernel_variance)}, shape: {kernel_variance.shape}"
            kernel_variance = kernel_variance[0]
        self._theta[self.D] = float(kernel_variance)

    def set_likelihood_variance(self, likelihood_variance):
        if isinstance(likelihood_variance, np.ndarray):
            assert (len(likelihood_variance) == 1) & (len(likelihood_variance.shape) == 1), \
                f"set_likelihood_variance expected to receive float, or np.array with len(1), shape:(1,), got" \
                f"len: {len(likelihood_variance)}, shape: {likelihood_variance.shape}"
            likelihood_variance = likelihood_variance[0]
        unconstrained = not np.isfinite(self._lo[self.D + 1])
        if unconstrained and likelihood_variance < LIKELIHOOD_VARIANCE_LOWER_BOUND:
            warnings.warn("\n***\ntrying to set likelihood_variance to value less than "
                          "model.likelihood.variance_lower_bound\nwill set to variance_lower_bound\n***\n")
            likelihood_variance = LIKELIHOOD_VARIANCE_LOWER_BOUND
        self._theta[self.D + 1] = float(likelihood_variance)

    # -- constraints: gpflow_models.py:416-590
    def _slice(self, name):
        D = self.D
        return {"lengthscales": slice(0, D), "kernel_variance": slice(D, D + 1),
                "likelihood_variance": slice(D + 1, D + 2)}[name]

    def _set_param_constraints(self, name, low, high, move_within_tol=True, tol=1e-8, scale=False,
                               scale_magnitude=None):
        """Box for one named parameter (behaviour of gpflow_models.py:416-494): bounds optionally divided by the
        coordinate scale, the current value pulled to at least ``tol`` inside the box; the sigmoid bijector itself
        runs on the GPU (lo / hi of the C ABI)."""
        sl = self._slice(name)
        n = sl.stop - sl.start
        bounds = []
        for label, b in (("low", low), ("high", high)):
            b = np.atleast_1d(np.asarray(b, dtype=np.float64))
            if b.ndim != 1:
                raise AssertionError(f"{label} constraint must be a scalar or 1-d")
            if len(b) != n:
                raise AssertionError(f"len of {label} constraint does not match param length")
            bounds.append(b)
        lo, hi = bounds
        if not np.all(lo <= hi):
            raise AssertionError("all values in high constraint must be greater than low")
        if scale:
            div = self.coords_scale[0, :] if scale_magnitude is None else scale_magnitude
            lo, hi = lo / div, hi / div
        cur = self._theta[sl]
        if move_within_tol:
            cur = clamp_within(cur, lo, hi, tol)
        self._theta[sl], self._lo[sl], self._hi[sl] = cur, lo, hi

    def set_lengthscales_constraints(self, low, high, move_within_tol=True, tol=1e-8, scale=False, scale_magnitude=None):
        self._set_param_constraints("lengthscales", low, high, move_within_tol, tol, scale, scale_magnitude)

    def set_kernel_variance_constraints(self, low, high, move_within_tol=True, tol=1e-8, scale=False, scale_magnitude=None):
        self._set_param_constraints("kernel_variance", low, high, move_within_tol, tol, scale, scale_magnitude)

    def set_likelihood_variance_constraints(self, low, high, move_within_tol=True, tol=1e-8, scale=False, scale_magnitude=None):
        self._set_param_constraints("likelihood_variance", low, high, move_within_tol, tol, scale, scale_magnitude)

    # -- the three device calls
    def _run(self, *, optimiser, max_iter=0, pred_coords=None, **opt_kwargs):
        N, D = self.coords.shape
        P = 0 if pred_coords is None else len(pred_coords)
        Xs = np.zeros((0, D)) if pred_coords is None else pred_coords
        return self._engine.fit_predict_batch(
            dtype=self.dtype, D=D, obs_off=np.array([0, N]), X=self.coords, y=self.obs[:, 0],
            pred_off=np.array([0, P]), Xs=Xs, theta0=self._theta[None, :], lo=self._lo[None, :],
            hi=self._hi[None, :], trainable=self._trainable, kernel=self.kernel, optimiser=optimiser,
            max_iter=max_iter, **opt_kwargs)

    def _fix_hyperparameters(self, params_list):
        # gpflow_models.py:275-288
        for param in params_list:
            if param in self.param_names:
                self._trainable[self._slice(param)] = False
            else:
                print(f"{param} is not detected as a hyperparameter. Skipping...")

    def optimise_parameters(self, max_iter=10_000, fixed_params=None, **opt_kwargs):
        """L-BFGS on the unconstrained parameters, entirely on the GPU
        (replaces gpflow.optimizers.Scipy().minimize, gpflow_models.py:291-329).
        Returns True when the optimiser converged within ``max_iter`` (scipy ``success``)."""
        if fixed_params is None:
            fixed_params = []
        self._fix_hyperparameters(fixed_params)
        optimiser = opt_kwargs.pop("optimiser", "lbfgs")
        # engine tolerances may be passed through; SciPy-specific keys of the reference are ignored
        known = {k: opt_kwargs[k] for k in ("max_ls", "ftol", "gtol", "adam_lr") if k in opt_kwargs}
        r = self._run(optimiser=optimiser, max_iter=max_iter, **known)
        self.status = int(r.status[0])
        self.n_eval = int(r.n_eval[0])
        if self.status in (0, 1, 6):
            self._theta = r.theta[0].copy()
        success = self.status == 0
        if not success:
            print("*" * 10)
            print("optimization failed!")
        return success

    def get_objective_function_value(self):
        """Negative log marginal likelihood at the current parameters (gpflow_models.py:334-337)."""
        r = self._run(optimiser="none")
        return float(r.nll[0])

    def predict(self, coords, full_cov=False, apply_scale=True) -> Dict[str, np.ndarray]:
        # gpflow_models.py:187-273
        if pd is not None and isinstance(coords, (pd.Series, pd.DataFrame)):
            if self.coords_col is not None:
                coords = coords[self.coords_col].values
            else:
                coords = coords.values
        if isinstance(coords, list):
            coords = np.array(coords)
        if len(coords.shape) == 1:
            coords = coords[None, :]
        assert isinstance(coords, np.ndarray), "coords should be an ndarray (one can be converted from)"
        coords = coords.astype(self.coords.dtype)
        if apply_scale:
            coords = coords / self.coords_scale
        r = self._run(optimiser="none", pred_coords=coords, full_cov=bool(full_cov))
        if r.status[0] in (2, 3):
            raise FloatingPointError("covariance matrix is not positive definite at the current parameters")
        if not full_cov:
            out = {"f*": r.f_mean.astype(np.float64), "f*_var": r.f_var.astype(np.float64),
                   "y_var": r.y_var.astype(np.float64)}
        else:
            # gpflow_models.py:245-263: marginal variance = diagonal of the full covariance; the predictive
            # covariance adds the likelihood variance on the diagonal
            P = len(coords)
            f_cov = np.asarray(r.f_cov, dtype=np.float64).reshape(P, P)
            f_var = np.diag(f_cov).copy()
            y_var = r.y_var.astype(np.float64)
            y_cov = f_cov.copy()
            y_cov[np.arange(P), np.arange(P)] += y_var - f_var
            out = {"f*": r.f_mean.astype(np.float64), "f*_var": f_var, "y_var": y_var, "f*_cov": f_cov, "y_cov": y_cov}
        f_bar = self.obs_mean[:, 0]
        if len(f_bar) != len(out["f*"]):
            assert len(f_bar) == 1, f"'f_bar' did not match the length of 'f*' and f_bar len is not, got: {len(f_bar)}"
            out["f_bar"] = np.repeat(f_bar, len(out["f*"]))
        else:
            out["f_bar"] = f_bar
        return out


def get_model(name):
    """Registry hook with the reference's semantics (GPSat/models/__init__.py:3-28): the exact-GP
    names resolve to the HIP backend; anything else is NotImplementedError."""
    if name in ("HipGPRModel", "GPflowGPRModel"):
        return HipGPRModel
    raise NotImplementedError(f"model with name: '{name}' is not implemented")
