"""Seeded synthetic expert tiles: no reference code, NumPy only.

Per tile t (seed = base_seed + t): X in R^{N x D} with x,y ~ U(-6,6), t ~ U(-4,4) in already-scaled
units; ground-truth lengthscales ~ U(2,8) (time: U(3,8)), kernel variance 0.015, likelihood variance
0.004; y ~ N(0, K + sn2 I) drawn in fp64 via Cholesky and de-meaned.  P prediction points uniformly
within radius 4 of the tile centre at t = 0.
"""
from __future__ import annotations

import numpy as np

_SPAN = np.array([6.0, 6.0, 4.0])


def _kernel(kid, X, X2, ell, sf2):
    A = X / ell
    B = X2 / ell
    r2 = np.maximum((A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T, 0.0)
    if kid == 0:
        return sf2 * np.exp(-0.5 * r2)
    r = np.sqrt(np.maximum(r2, 1e-36))
    if kid == 1:
        return sf2 * np.exp(-r)
    if kid == 2:
        s = np.sqrt(3.0) * r
        return sf2 * (1 + s) * np.exp(-s)
    s = np.sqrt(5.0) * r
    return sf2 * (1 + s + s * s / 3.0) * np.exp(-s)


def make_tile(seed, N, P, D=3, kid=0, sf2=0.015, sn2=0.004):
    rng = np.random.default_rng(seed)
    span = _SPAN[:D] if D <= 3 else np.full(D, 6.0)
    X = rng.uniform(-1.0, 1.0, (N, D)) * span
    ell = np.array([rng.uniform(2, 8) if d < 2 else rng.uniform(3, 8) for d in range(D)])
    if N > 0:
        K = _kernel(kid, X, X, ell, sf2) + sn2 * np.eye(N)
        y = np.linalg.cholesky(K) @ rng.standard_normal(N)
        y = y - y.mean()
    else:
        y = np.zeros(0)
    ang = rng.uniform(0, 2 * np.pi, P)
    rad = 4.0 * np.sqrt(rng.uniform(0, 1, P))
    Xs = np.zeros((P, D))
    Xs[:, 0] = rad * np.cos(ang)
    if D > 1:
        Xs[:, 1] = rad * np.sin(ang)
    return X, y, Xs, np.concatenate([ell, [sf2, sn2]])


def make_batch(T, N, P, D=3, kid=0, base_seed=0, dtype=np.float32):
    """N, P: int or per-tile sequences.  Returns a dict with the packed ragged (CSR) arrays."""
    Ns = np.broadcast_to(np.asarray(N), (T,)).astype(np.int64)
    Ps = np.broadcast_to(np.asarray(P), (T,)).astype(np.int64)
    obs_off = np.concatenate([[0], np.cumsum(Ns)])
    pred_off = np.concatenate([[0], np.cumsum(Ps)])
    X = np.empty((obs_off[-1], D))
    y = np.empty(obs_off[-1])
    Xs = np.empty((pred_off[-1], D))
    truth = np.empty((T, D + 2))
    for t in range(T):
        x_, y_, xs_, th = make_tile(base_seed + t, int(Ns[t]), int(Ps[t]), D, kid)
        X[obs_off[t]:obs_off[t + 1]] = x_
        y[obs_off[t]:obs_off[t + 1]] = y_
        Xs[pred_off[t]:pred_off[t + 1]] = xs_
        truth[t] = th
    return dict(T=T, D=D, kid=kid, obs_off=obs_off, pred_off=pred_off, X=X.astype(dtype), y=y.astype(dtype),
                Xs=Xs.astype(dtype), truth=truth)


def default_bounds(T, D):
    """Lengthscale box [1e-8, (12,12,9)] as in configs/example_local_expert_oi.json:106-126 after scaling;
    variances unconstrained (softplus)."""
    hi_l = np.array([12.0, 12.0, 9.0])[:D] if D <= 3 else np.full(D, 12.0)
    lo = np.full((T, D + 2), np.nan)
    hi = np.full((T, D + 2), np.nan)
    lo[:, :D] = 1e-8
    hi[:, :D] = hi_l
    return lo, hi
