"""GPU parity of the post-processing kernels (gpsat_post.hip through the C ABI) against the reference's own outputs
(tests/golden/ref_post.npz) and the oracle: fp64, tolerance 1e-12 relative (only the summation order differs; 1e-14 absolute where a
weighted sum of signed values cancels), NaN pattern identical."""
import os

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_post.npz"))
RTOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    from gpsat_amd.engine import default_engine
    return default_engine()


def test_smooth_matches_reference(eng):
    out = eng.smooth_batch(G["sx"], G["sy"], G["svals"], float(G["lx"]), float(G["ly"]))
    np.testing.assert_allclose(out, G["smoothed"], rtol=RTOL)
    tiny = eng.smooth_batch(G["sx"], G["sy"], G["svals"], 1.0, 1.0)
    np.testing.assert_array_equal(np.isnan(tiny), np.isnan(G["smoothed_tiny"]))
    ok = ~np.isnan(tiny)
    np.testing.assert_allclose(tiny[ok], G["smoothed_tiny"][ok], rtol=RTOL)
    assert np.isnan(eng.smooth_batch(G["sx"], G["sy"], np.full_like(G["svals"], np.nan), 1e5, 1e5)).all()
    assert len(eng.smooth_batch(np.zeros(0), np.zeros(0), np.zeros(0), 1.0, 1.0)) == 0


@pytest.mark.parametrize("T", [1, 63, 64, 65, 1000, 20000])
def test_smooth_matches_oracle_sizes(eng, T):
    from oracle import post_oracle as po
    rng = np.random.default_rng(T)
    x, y = rng.uniform(-1, 1, T), rng.uniform(-1, 1, T)
    v = rng.standard_normal(T)
    v[rng.random(T) < 0.1] = np.nan
    out = eng.smooth_batch(x, y, v, 0.2, 0.1)
    n = min(T, 200)                                   # the sequential oracle loop is O(T^2) in Python
    ref = po.gaussian_2d_weight(x[:n], y[:n], x, y, 0.2, 0.1, v)
    np.testing.assert_allclose(out[:n], ref, rtol=1e-11, atol=1e-14)
    # properties at full size: constant field is reproduced; output within [min, max] of the inputs
    c = eng.smooth_batch(x, y, np.full(T, 3.25), 0.2, 0.1)
    np.testing.assert_allclose(c, 3.25, rtol=1e-14)
    ok = ~np.isnan(out)
    if ok.any():
        assert np.nanmin(v) - 1e-12 <= out[ok].min() and out[ok].max() <= np.nanmax(v) + 1e-12
    assert np.array_equal(out, eng.smooth_batch(x, y, v, 0.2, 0.1), equal_nan=True)      # reproducible


def test_smooth_hyperparameters_tables(eng, tmp_path):
    """table-level function: clipping, per-slice smoothing over the other dims and _dim_*, NaN rows dropped,
    unsmoothed parameters copied (GPSat/postprocessing.py:215-343)."""
    from gpsat_amd.postprocessing import smooth_hyperparameters
    from gpsat_amd.local_experts import ResultStore
    from oracle import post_oracle as po
    rng = np.random.default_rng(5)
    xs, ys, ts = np.meshgrid(np.arange(5) * 1e5, np.arange(4) * 1e5, [10.0, 11.0])
    locs = pd.DataFrame({"x": xs.ravel(), "y": ys.ravel(), "t": ts.ravel()})
    cc = ["x", "y", "t"]
    ls = pd.concat([locs.assign(_dim_0=d, lengthscales=rng.uniform(0.5, 20, len(locs))) for d in range(3)])
    ls.loc[ls.index[3], "lengthscales"] = np.nan
    kv = locs.assign(_dim_0=0, kernel_variance=rng.uniform(0.1, 2, len(locs)))
    store = ResultStore(str(tmp_path / "res"))
    store.append("lengthscales", ls.set_index(cc))
    store.append("kernel_variance", kv.set_index(cc))
    cfg = {"lengthscales": {"l_x": 2e5, "l_y": 1.5e5, "max": 12.0, "min": 1.0}}
    out = smooth_hyperparameters(str(tmp_path / "res"), ["lengthscales"], cfg, xy_dims=["x", "y"], engine=eng)
    assert set(out) == {"lengthscales_SMOOTHED", "kernel_variance_SMOOTHED"}
    pd.testing.assert_frame_equal(out["kernel_variance_SMOOTHED"], kv.set_index(cc))
    sm = out["lengthscales_SMOOTHED"].reset_index()
    assert list(sm.columns) == list(ls.columns) and len(sm) == len(ls)
    for (t, d), sub in ls.groupby(["t", "_dim_0"]):
        v = np.clip(sub["lengthscales"].values, 1.0, 12.0)
        ref = po.gaussian_2d_weight(sub["x"].values, sub["y"].values, sub["x"].values, sub["y"].values, 2e5, 1.5e5, v)
        got = sm[(sm["t"] == t) & (sm["_dim_0"] == d)]
        np.testing.assert_array_equal(got[["x", "y"]].values, sub[["x", "y"]].values)
        np.testing.assert_allclose(got["lengthscales"].values, ref, rtol=RTOL)
    # written to the store under the new names; usable by load_params
    assert ResultStore(str(tmp_path / "res")).read("lengthscales_SMOOTHED") is not None
    with pytest.raises(NotImplementedError):
        smooth_hyperparameters(str(tmp_path / "res"), ["likelihood_variance"], cfg, engine=eng)


def test_glue_matches_reference(eng):
    from gpsat_amd.postprocessing import glue_local_predictions_1d, glue_local_predictions_2d
    p1 = pd.DataFrame(G["p1"], columns=["x", "pred_loc_x", "f*", "f*_var"])
    g1 = glue_local_predictions_1d(p1, "pred_loc_x", "x", ["f*", "f*_var"], float(G["r1"]), engine=eng)
    assert list(g1.columns) == ["pred_loc_x", "f*", "f*_var"]
    np.testing.assert_allclose(g1.values, G["g1"], rtol=RTOL, atol=1e-14)
    g1s = glue_local_predictions_1d(p1, "pred_loc_x", "x", "f*", float(G["r1"]), engine=eng)
    np.testing.assert_allclose(g1s["f*"].values, G["g1"][:, 1], rtol=RTOL)
    p2 = pd.DataFrame(G["p2"], columns=["x", "y", "pred_loc_x", "pred_loc_y", "f*", "f*_var", "y_var"])
    g2 = glue_local_predictions_2d(p2, ["pred_loc_x", "pred_loc_y"], ["x", "y"], ["f*", "f*_var", "y_var"], float(G["r2"]),
                                   engine=eng)
    np.testing.assert_allclose(g2.values, G["g2"], rtol=RTOL, atol=1e-14)


def test_glue_radius_dict_and_errors(eng):
    from gpsat_amd.postprocessing import glue_local_predictions_1d
    from gpsat_amd.engine import GpsatError
    from oracle import post_oracle as po
    p1 = pd.DataFrame(G["p1"], columns=["x", "pred_loc_x", "f*", "f*_var"])
    radii = {loc: 0.1 + 0.02 * i for i, loc in enumerate(np.unique(p1["x"].values))}
    got = glue_local_predictions_1d(p1, "pred_loc_x", "x", ["f*", "f*_var"], radii, engine=eng)
    ref = po.glue_local_predictions(p1, "pred_loc_x", "x", ["f*", "f*_var"], radii)
    np.testing.assert_allclose(got.values, ref.values, rtol=RTOL, atol=1e-14)
    with pytest.raises(TypeError):
        glue_local_predictions_1d(p1, "pred_loc_x", "x", "f*", "0.1", engine=eng)
    with pytest.raises(ValueError):
        glue_local_predictions_1d(p1.assign(a=1, b=2, c=3), "pred_loc_x", "x", ["f*", "f*_var", "a", "b", "c"], 0.1, engine=eng)
    with pytest.raises(GpsatError):
        eng.glue_batch(np.array([0, 2]), np.zeros((1, 3)), np.zeros((1, 3)), np.zeros((1, 3)), 1.0)   # seg != R


def test_glue_large_properties(eng):
    """2M rows / 500k locations: single-expert locations pass through unchanged; identical predictions glue to
    themselves; result independent of the row order."""
    from gpsat_amd.postprocessing import glue_local_predictions_2d
    rng = np.random.default_rng(9)
    L = 500_000
    px, py = rng.integers(0, 4000, L).astype(float), rng.integers(0, 4000, L).astype(float)
    k = rng.integers(1, 8, L)
    rep = np.repeat(np.arange(L), k)
    df = pd.DataFrame({"pred_loc_x": px[rep], "pred_loc_y": py[rep]})
    df["x"] = df["pred_loc_x"] + rng.uniform(-300, 300, len(df))
    df["y"] = df["pred_loc_y"] + rng.uniform(-300, 300, len(df))
    df["f*"] = np.sin(df["pred_loc_x"] * 1e-3) + 0.0 * df["x"]
    df["v"] = rng.uniform(0.5, 1.5, len(df))
    g = glue_local_predictions_2d(df, ["pred_loc_x", "pred_loc_y"], ["x", "y"], ["f*", "v"], 300.0, engine=eng)
    np.testing.assert_allclose(g["f*"].values, np.sin(g["pred_loc_x"].values * 1e-3), rtol=1e-12, atol=1e-14)
    lo = df.groupby(["pred_loc_x", "pred_loc_y"])["v"].agg(["min", "max"]).reset_index()
    assert (g["v"].values >= lo["min"].values - 1e-12).all() and (g["v"].values <= lo["max"].values + 1e-12).all()
    g2 = glue_local_predictions_2d(df.sample(frac=1.0, random_state=1), ["pred_loc_x", "pred_loc_y"], ["x", "y"],
                                   ["f*", "v"], 300.0, engine=eng)
    np.testing.assert_allclose(g2.values, g.values, rtol=1e-12, atol=1e-14)
