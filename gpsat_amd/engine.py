"""Batched local-expert GP engine: thin host wrapper over the C ABI (include/gpsat_hip.h).

``Engine.fit_predict_batch`` is the packed-ragged counterpart of the per-tile body of
LocalExpertOI.run (GPSat/local_experts.py:1043-1159): all tiles of a wave are fitted and
predicted by ONE call / one kernel launch.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


class GpsatError(RuntimeError):
    pass


@dataclass
class BatchResult:
    theta: np.ndarray      # [T, H] learned parameters (l_1..l_D, kernel_variance, likelihood_variance)
    nll: np.ndarray        # [T] objective = negative log marginal likelihood
    status: np.ndarray     # [T] see _lib.STATUS
    n_eval: np.ndarray     # [T] objective+gradient evaluations used by the optimiser
    f_mean: object         # [sum P] numpy (host mode) or torch tensor (device mode)
    f_var: object
    y_var: object
    grad: np.ndarray | None = None   # [T, H] dNLL/dtheta at theta (when requested)
    f_cov: object = None   # full_cov: flat [sum P_t^2] (numpy / torch as f_mean); tile t = f_cov[cov_off[t]:cov_off[t+1]].reshape(P_t, P_t)
    cov_off: np.ndarray | None = None
    n_iter: np.ndarray | None = None   # [T] optimiser iterations completed (scipy nit)
    kernel_ms: float = 0.0
    total_ms: float = 0.0


def centre_tiles(X, Xs, obs_off, pred_off):
    """Subtract every tile's mean coordinate from its observations and prediction points (fp64).

    The covariance functions are stationary, so the model is unchanged; what changes is the rounding of the cast to
    fp32 that follows: GPSat coordinates are typically far from the origin (t ~ 18 000 days against length scales of a
    few days), where fp32 resolves the scaled coordinate to ~1e-3 only.  Centred, the cast error is relative to the
    tile's own extent."""
    Ns, Ps = np.diff(obs_off), np.diff(pred_off)
    nz = Ns > 0
    c = np.zeros((len(Ns), X.shape[1]))
    if nz.any():
        c[nz] = np.add.reduceat(X, obs_off[:-1][nz], axis=0) / Ns[nz, None]
    return X - np.repeat(c, Ns, axis=0), Xs - np.repeat(c, Ps, axis=0)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """One engine per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device_id: int = 0, workgroups_per_cu: int = 0):
        self._lib = L.get_lib()
        opts = L.GpsatOpts()
        opts.workgroups_per_cu = int(workgroups_per_cu)
        h = C.c_void_p()
        rc = self._lib.gpsat_create(int(device_id), C.byref(opts), C.byref(h))
        if rc != 0:
            raise GpsatError(f"gpsat_create failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        self._h = h
        buf = C.create_string_buffer(256)
        self._lib.gpsat_device_name(self._h, buf, 256)
        self.device_name = buf.value.decode()
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpsat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fit_predict_batch(self, *, D, obs_off, X, y, pred_off, Xs, theta0, lo=None, hi=None,
                          trainable=None, kernel="Matern32", optimiser="lbfgs", max_iter=10_000,
                          max_ls=0, ftol=0.0, gtol=0.0, adam_lr=0.0, want_grad=False,
                          out=None, dtype="f32", full_cov=False) -> BatchResult:
        """
        X [sumN, D], y [sumN], Xs [sumP, D]: numpy arrays (host mode) or contiguous torch.cuda tensors (device
        mode; outputs are then torch tensors, optionally preallocated via ``out`` = (f_mean, f_var, y_var)).
        ``dtype``: "f32" (default; fp32 MFMA kernels) or "f64" (the reference's native precision, fp64 MFMA
        kernels); host arrays are cast, device tensors must already have that dtype.  Offsets / theta0 / bounds
        are always host numpy (fp64).  ``full_cov``: also return the P_t x P_t posterior covariance of every tile
        (predict(full_cov=True), gpflow_models.py:245-263).
        """
        obs_off = np.ascontiguousarray(obs_off, dtype=np.int64)
        pred_off = np.ascontiguousarray(pred_off, dtype=np.int64)
        T = len(obs_off) - 1
        H = D + 2
        assert len(pred_off) == T + 1
        theta0 = np.ascontiguousarray(np.broadcast_to(np.asarray(theta0, dtype=np.float64), (T, H)))
        lo = np.full((T, H), np.nan) if lo is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(lo, dtype=np.float64), (T, H)))
        hi = np.full((T, H), np.nan) if hi is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(hi, dtype=np.float64), (T, H)))
        trainable = np.ones(H, dtype=np.uint8) if trainable is None else \
            np.ascontiguousarray(np.asarray(trainable).astype(bool).astype(np.uint8))
        assert trainable.shape == (H,)
        sumN, sumP = int(obs_off[-1]), int(pred_off[-1])

        if dtype not in ("f32", "f64"):
            raise GpsatError(f"dtype {dtype!r}: use 'f32' or 'f64'")
        np_dt = np.float32 if dtype == "f32" else np.float64
        device_mode = not isinstance(X, np.ndarray)
        if device_mode:
            import torch
            t_dt = torch.float32 if dtype == "f32" else torch.float64
            for tname, t_ in (("X", X), ("y", y), ("Xs", Xs)):
                if not (isinstance(t_, torch.Tensor) and t_.is_cuda and t_.dtype == t_dt and t_.is_contiguous()):
                    raise GpsatError(f"{tname}: device mode needs contiguous {dtype} CUDA tensors")
            if X.device.index != self.device_id:
                raise GpsatError(f"tensors live on cuda:{X.device.index}, engine on device {self.device_id}")
            assert X.numel() == sumN * D and y.numel() == sumN and Xs.numel() == sumP * D
            if out is None:
                fm = torch.empty(max(sumP, 1), dtype=t_dt, device=X.device)
                fv = torch.empty_like(fm)
                yv = torch.empty_like(fm)
            else:
                fm, fv, yv = out
            torch.cuda.current_stream(X.device).synchronize()   # inputs must be complete before our stream reads them
            pX, py, pXs = X.data_ptr(), y.data_ptr(), Xs.data_ptr()
            pfm, pfv, pyv = fm.data_ptr(), fv.data_ptr(), yv.data_ptr()
        else:
            if dtype == "f32" and sumN > 0 and np.asarray(X).dtype == np.float64:
                # fp64 coordinates handed to the fp32 kernels: centre per tile before the cast (see centre_tiles)
                X, Xs = centre_tiles(np.asarray(X, dtype=np.float64).reshape(sumN, D),
                                     np.asarray(Xs, dtype=np.float64).reshape(sumP, D), obs_off, pred_off)
            X = np.ascontiguousarray(X, dtype=np_dt).reshape(sumN, D)
            y = np.ascontiguousarray(y, dtype=np_dt).reshape(sumN)
            Xs = np.ascontiguousarray(Xs, dtype=np_dt).reshape(sumP, D)
            fm = np.empty(sumP, dtype=np_dt)
            fv = np.empty(sumP, dtype=np_dt)
            yv = np.empty(sumP, dtype=np_dt)
            pX, py, pXs = _ptr(X), _ptr(y), _ptr(Xs)
            pfm, pfv, pyv = _ptr(fm), _ptr(fv), _ptr(yv)

        cov_off = fc = pfc = None
        if full_cov:
            Pt = np.diff(pred_off)
            cov_off = np.concatenate([[0], np.cumsum(Pt * Pt)]).astype(np.int64)
            if device_mode:
                fc = torch.empty(max(int(cov_off[-1]), 1), dtype=t_dt, device=X.device)
                pfc = fc.data_ptr()
            else:
                fc = np.empty(max(int(cov_off[-1]), 1), dtype=np_dt)
                pfc = _ptr(fc)
        theta = np.empty((T, H), dtype=np.float64)
        nll = np.empty(T, dtype=np.float64)
        grad = np.empty((T, H), dtype=np.float64) if want_grad else None
        status = np.empty(T, dtype=np.int32)
        n_eval = np.empty(T, dtype=np.int32)
        n_iter = np.zeros(T, dtype=np.int32)

        b = L.GpsatBatch()
        b.T, b.D, b.dtype = T, D, (L.F32 if dtype == "f32" else L.F64)
        b.kernel = L.KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        b.memory = L.MEM_DEVICE if device_mode else L.MEM_HOST
        b.optimiser = L.OPT_IDS[optimiser] if not isinstance(optimiser, int) else optimiser
        b.max_iter, b.max_ls = int(max_iter), int(max_ls)
        b.ftol, b.gtol, b.adam_lr = float(ftol), float(gtol), float(adam_lr)
        b.obs_off, b.pred_off = _ptr(obs_off), _ptr(pred_off)
        b.theta0, b.lo, b.hi, b.trainable = _ptr(theta0), _ptr(lo), _ptr(hi), _ptr(trainable)
        b.X, b.y, b.Xs = pX, py, pXs
        b.theta, b.nll, b.grad = _ptr(theta), _ptr(nll), _ptr(grad)
        b.status, b.n_eval, b.n_iter = _ptr(status), _ptr(n_eval), _ptr(n_iter)
        b.f_mean, b.f_var, b.y_var = pfm, pfv, pyv
        b.cov_off, b.f_cov = (_ptr(cov_off), pfc) if full_cov else (None, None)
        rc = self._lib.gpsat_fit_predict_batch(self._h, C.byref(b))
        if rc != 0:
            raise GpsatError(f"gpsat_fit_predict_batch failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        km, tm = C.c_double(), C.c_double()
        self._lib.gpsat_last_timing(self._h, C.byref(km), C.byref(tm))
        if device_mode:
            fm, fv, yv = fm[:sumP], fv[:sumP], yv[:sumP]
        return BatchResult(theta=theta, nll=nll, status=status, n_eval=n_eval, n_iter=n_iter, f_mean=fm, f_var=fv, y_var=yv,
                           grad=grad, kernel_ms=km.value, total_ms=tm.value,
                           f_cov=(fc[:int(cov_off[-1])] if full_cov else None), cov_off=cov_off)


    def select_batch(self, points: np.ndarray, refs: np.ndarray, criteria, points_cm: np.ndarray = None):
        """Batched tile selection on the GPU (gpsat_select_batch).

        points [M, C] fp64, refs [T, C] fp64 (same column numbering); criteria: list of
        ("cmp", col, comp, val)  ->  points[:, col] <comp> refs[:, col] + val
        ("ball", [cols], comp, r) -> Euclidean ball, comp "<=" (inclusive) or "<" (strict).
        Returns (off [T+1] int64, idx [off[-1]] int32): selected rows per expert in source order."""
        refs = np.ascontiguousarray(refs, dtype=np.float64)
        T = refs.shape[0]
        if points_cm is not None:                                      # the table already column-major [C][M] (kept by the caller)
            pts_cm = np.ascontiguousarray(points_cm, dtype=np.float64)
            Cc, M = pts_cm.shape
        else:
            points = np.asarray(points, dtype=np.float64)
            M, Cc = points.shape
            pts_cm = np.ascontiguousarray(points.T)                   # column-major [C][M]
        assert refs.shape[1] == Cc
        sp = L.GpsatSelectSpec()
        if not 1 <= len(criteria) <= L.SEL_MAXCRIT:
            raise GpsatError(f"1..{L.SEL_MAXCRIT} criteria supported")
        sp.n_crit = len(criteria)
        for k, (kind, cols, comp, val) in enumerate(criteria):
            sp.kind[k] = 0 if kind == "cmp" else 1
            sp.comp[k] = L.COMP_IDS[comp]
            cl = [cols] if kind == "cmp" else list(cols)
            sp.ncols[k] = len(cl)
            for m_, c_ in enumerate(cl):
                sp.cols[k][m_] = int(c_)
            sp.val[k] = float(val)
        off = np.zeros(T + 1, dtype=np.int64)
        rc = self._lib.gpsat_select_batch(self._h, C.byref(sp), M, Cc, _ptr(pts_cm), T, _ptr(refs), _ptr(off), None, 0)
        if rc != 0:
            raise GpsatError(f"gpsat_select_batch failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        idx = np.empty(int(off[-1]), dtype=np.int32)
        if len(idx):
            rc = self._lib.gpsat_select_batch(self._h, C.byref(sp), M, Cc, _ptr(pts_cm), T, _ptr(refs), _ptr(off),
                                              _ptr(idx), len(idx))
            if rc != 0:
                raise GpsatError(f"gpsat_select_batch failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        return off, idx

    def smooth_batch(self, x, y, vals, l_x: float, l_y: float) -> np.ndarray:
        """Gaussian smoothing of one hyper-parameter field on the GPU (gpsat_smooth_batch; replaces
        gaussian_2d_weight, GPSat/postprocessing.py:22-52, with x0, y0 = x, y as smooth_hyperparameters calls it)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        assert x.shape == y.shape == vals.shape and x.ndim == 1
        out = np.empty_like(vals)
        rc = self._lib.gpsat_smooth_batch(self._h, len(x), _ptr(x), _ptr(y), _ptr(vals), float(l_x), float(l_y), _ptr(out))
        if rc != 0:
            raise GpsatError(f"gpsat_smooth_batch failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        return out

    def glue_batch(self, seg, pred, xprt, vals, sigma) -> np.ndarray:
        """Weighted combination of overlapping predictions (gpsat_glue_batch): rows already sorted into segments
        seg [G+1]; pred, xprt [ndim, R]; vals [nvars, R]; sigma scalar or per row [R]; returns [nvars, G]."""
        seg = np.ascontiguousarray(seg, dtype=np.int64)
        pred = np.ascontiguousarray(pred, dtype=np.float64)
        xprt = np.ascontiguousarray(xprt, dtype=np.float64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        ndim, R = pred.shape
        nvars = vals.shape[0]
        assert xprt.shape == pred.shape and vals.shape[1] == R
        G = len(seg) - 1
        out = np.empty((nvars, G), dtype=np.float64)
        srow = None
        if np.ndim(sigma) > 0:
            srow = np.ascontiguousarray(sigma, dtype=np.float64)
            assert srow.shape == (R,)
        rc = self._lib.gpsat_glue_batch(self._h, R, G, ndim, nvars, _ptr(seg), _ptr(pred), _ptr(xprt), _ptr(vals),
                                        0.0 if srow is not None else float(sigma), _ptr(srow) if srow is not None else None,
                                        _ptr(out))
        if rc != 0:
            raise GpsatError(f"gpsat_glue_batch failed ({rc}): {self._lib.gpsat_last_error().decode()}")
        return out


_default_engine = None


def default_engine() -> Engine:
    """Process-wide engine on LOCAL_RANK's GPU (one process per GPU)."""
    global _default_engine
    if _default_engine is None:
        import os
        _default_engine = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_engine
