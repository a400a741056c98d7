"""GPU: repeated launches return the same BITS -- objective, gradient and predictions of every tile -- in both fp32 builds.

Why this is a test of its own: the K^-1 phase of the 8-wave build multiplies its blocks as exact
three-plane bf16 products (`v_mfma_f32_32x32x16_bf16`).  The same loop in the 4-wave build, where two workgroups share a CU,
made the OTHER workgroup's fp32 factorisation come out different in the last bit from launch to launch (11-420 of 4096 tiles,
objective / gradient / predictions together) -- inside every parity tolerance of the suite, invisible to it, and fatal to
what the time-sliced queue, the cooperative tiles and the store's resume promise.  Enough tiles that every CU holds two
workgroups of the 4-wave build."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from gpsat_amd import synthetic as syn

pytestmark = pytest.mark.gpu

LAUNCHES = 4


def _batch(T, N, P, D, kid):
    from threadpoolctl import threadpool_limits
    with threadpool_limits(1):
        with ThreadPoolExecutor(16) as pool:                  # threads, not processes: this process may already hold the GPU
            tiles = list(pool.map(lambda t: syn.make_tile(7_000_000 + t, N, P, D, kid), range(T)))
    X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
    y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
    Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
    th = np.exp(np.random.default_rng(11).normal(0.0, 0.5, (T, D + 2)))
    return dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, X=X, y=y, pred_off=np.arange(T + 1, dtype=np.int64) * P, Xs=Xs,
                theta0=th, kernel=kid, optimiser="none", want_grad=True)


# (the 4-wave build runs tiles of up to 832 points -- 80 KiB of LDS per workgroup --, the 8-wave build the larger ones)
@pytest.mark.parametrize("wg_per_cu, T, N", [(0, 1536, 500), (0, 768, 700), (1, 512, 500), (0, 320, 1000)],
                         ids=["4-wave-build-two-workgroups-per-cu", "4-wave-build-700-points", "8-wave-build-forced",
                              "8-wave-build-large-tiles"])
def test_repeated_launches_are_bit_identical(wg_per_cu, T, N):
    from gpsat_amd.engine import Engine
    kw = _batch(T, N, 8, 3, 0)
    eng = Engine(0, workgroups_per_cu=wg_per_cu)
    try:
        runs = [eng.fit_predict_batch(**kw) for _ in range(LAUNCHES)]
    finally:
        eng.close()
    r0 = runs[0]
    assert np.isfinite(r0.nll).all() and np.isfinite(r0.grad).all()
    for k, r in enumerate(runs[1:], 1):
        bad = np.nonzero((r.nll != r0.nll) | (r.grad != r0.grad).any(axis=1)
                         | (r.f_mean != r0.f_mean).reshape(T, -1).any(axis=1) | (r.f_var != r0.f_var).reshape(T, -1).any(axis=1))[0]
        assert bad.size == 0, f"launch {k}: {bad.size} of {T} tiles differ from launch 0 (first: {bad[:8].tolist()})"
