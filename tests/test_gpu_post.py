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


def test_glue_skips_nan_predictions_like_pandas(eng):
    """A failed tile writes NaN predictions; the reference's groupby(...).sum() leaves NaN out of the weighted sum and
    keeps the row's weight in the denominator (GPSat/postprocessing.py:512-520) -- one failed expert must not turn every
    location it overlaps into NaN."""
    from gpsat_amd.postprocessing import glue_local_predictions_1d
    from oracle import post_oracle as po
    p1 = pd.DataFrame(G["p1"], columns=["x", "pred_loc_x", "f*", "f*_var"]).copy()
    bad = p1["x"] == np.unique(p1["x"].values)[1]                    # every prediction of one expert
    p1.loc[bad, ["f*", "f*_var"]] = np.nan
    got = glue_local_predictions_1d(p1, "pred_loc_x", "x", ["f*", "f*_var"], float(G["r1"]), engine=eng)
    ref = po.glue_local_predictions(p1, "pred_loc_x", "x", ["f*", "f*_var"], float(G["r1"]))
    assert np.isfinite(ref["f*"].values).all()
    np.testing.assert_allclose(got.values, ref.values, rtol=RTOL, atol=1e-14)


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


def test_fit_smooth_predict_glue_production_loop(eng, tmp_path):
    """The production loop either side of the GP engine: fit on a grid of experts -> smooth the
    hyper-parameter fields on the device -> predict-only re-run with the smoothed parameters loaded per tile
    (optimise=False) -> glue the overlapping predictions on the device.  Checks the plumbing (tables, suffixes, loaded
    values) and that the glued field reproduces the truth."""
    from gpsat_amd.local_experts import BatchedLocalExpertOI, ResultStore
    from gpsat_amd.postprocessing import smooth_hyperparameters, glue_local_predictions_2d
    rng = np.random.default_rng(0)
    M = 6000
    xy = rng.uniform(-1.0, 1.0, (M, 2))
    truth = lambda x, y: np.sin(3 * x) * np.cos(2 * y)
    df = pd.DataFrame({"x": xy[:, 0], "y": xy[:, 1], "obs": truth(xy[:, 0], xy[:, 1]) + 0.05 * rng.standard_normal(M)})
    gx, gy = np.meshgrid(np.linspace(-0.6, 0.6, 4), np.linspace(-0.6, 0.6, 4))
    xprt = pd.DataFrame({"x": gx.ravel(), "y": gy.ravel()})
    px, py = np.meshgrid(np.linspace(-0.7, 0.7, 29), np.linspace(-0.7, 0.7, 29))
    pred_df = pd.DataFrame({"x": px.ravel(), "y": py.ravel()})
    r_train, r_pred = 0.35, 0.3
    common = dict(
        expert_loc_config={"source": xprt},
        data_config={"data_source": df, "obs_col": ["obs"], "coords_col": ["x", "y"],
                     "local_select": [{"col": ["x", "y"], "comp": "<", "val": r_train}]},
        pred_loc_config={"method": "from_dataframe", "df": pred_df, "max_dist": r_pred}, engine=eng)
    model = {"oi_model": "HipGPRModel", "init_params": {"kernel": "Matern32", "obs_mean": "local"},
             "constraints": {"lengthscales": {"low": [1e-3, 1e-3], "high": [5.0, 5.0]}}}
    store = str(tmp_path / "store")
    tabs = BatchedLocalExpertOI(model_config=model, **common).run(store_path=store)
    assert len(tabs["run_details"]) == 16 and tabs["run_details"]["num_obs"].min() > 300
    # smooth all three hyper-parameter fields; values are clipped first
    cfg = {"lengthscales": {"l_x": 0.5, "l_y": 0.5, "max": 4.0}, "kernel_variance": {"l_x": 0.5, "l_y": 0.5},
           "likelihood_variance": {"l_x": 0.5, "l_y": 0.5, "min": 1e-4}}
    sm = smooth_hyperparameters(store, list(cfg), cfg, xy_dims=["x", "y"], engine=eng)
    ls, ls_s = tabs["lengthscales"], sm["lengthscales_SMOOTHED"]
    assert len(ls_s) == len(ls) and ls_s["lengthscales"].std() < ls["lengthscales"].std()
    assert ResultStore(store).read("kernel_variance_SMOOTHED") is not None
    # the derived predict-only configuration of the next step (GPSat/postprocessing.py:350-380)
    import json
    derived = json.load(open(sm["__config_file__"]))
    assert derived[-1]["run_kwargs"]["optimise"] is False and derived[-1]["run_kwargs"]["table_suffix"] == "_SMOOTHED"
    assert derived[-1]["model"]["load_params"] == {"file": store, "table_suffix": "_SMOOTHED"}
    assert derived[-1]["model"]["init_params"]["kernel"] == "Matern32"
    # predict-only with the smoothed parameters loaded per expert location
    model2 = dict(model, load_params=derived[-1]["model"]["load_params"])
    tabs2 = BatchedLocalExpertOI(model_config=model2, **common).run(store_path=store, optimise=False, table_suffix="_SMOOTHED")
    rd = tabs2["run_details_SMOOTHED"] if "run_details_SMOOTHED" in tabs2 else tabs2["run_details"]
    assert len(rd) == 16 and not rd["optimise_success"].any()                   # optimise=False => success False
    # loading from the same store + suffix without optimising: the parameter tables are NOT re-written
    # (local_experts.py:1090-1097); the smoothed tables stay as they were
    assert len(tabs2["lengthscales_SMOOTHED"]) == 0
    pd.testing.assert_frame_equal(ResultStore(store).read("lengthscales_SMOOTHED"), ls_s)
    # the loaded values were used: the objective of expert 5 equals that of a stand-alone model with the smoothed values
    from gpsat_amd.models import HipGPRModel
    loc = xprt.iloc[5]
    sel = df[(df["x"] - loc["x"]) ** 2 + (df["y"] - loc["y"]) ** 2 <= r_train * r_train]
    m = HipGPRModel(data=sel, obs_col="obs", coords_col=["x", "y"], obs_mean="local", kernel="Matern32", engine=eng)
    key = (loc["x"], loc["y"])
    m.set_parameters(lengthscales=ls_s.loc[[key]].sort_values("_dim_0")["lengthscales"].values,
                     kernel_variance=float(sm["kernel_variance_SMOOTHED"].loc[[key]]["kernel_variance"].values[0]),
                     likelihood_variance=float(sm["likelihood_variance_SMOOTHED"].loc[[key]]["likelihood_variance"].values[0]))
    assert rd.loc[[key]]["num_obs"].values[0] == len(sel)
    assert m.get_objective_function_value() == pytest.approx(rd.loc[[key]]["objective_value"].values[0], rel=1e-5, abs=1e-3)
    preds = (tabs2["preds_SMOOTHED"] if "preds_SMOOTHED" in tabs2 else tabs2["preds"]).reset_index()
    preds["f"] = preds["f*"] + preds["f_bar"]
    glued = glue_local_predictions_2d(preds, ["pred_loc_x", "pred_loc_y"], ["x", "y"], ["f", "f*_var"], r_pred, engine=eng)
    assert len(glued) == len(preds.drop_duplicates(["pred_loc_x", "pred_loc_y"]))
    err = glued["f"].values - truth(glued["pred_loc_x"].values, glued["pred_loc_y"].values)
    assert np.sqrt(np.mean(err ** 2)) < 0.03                                        # noise 0.05, ~700 obs per expert
    assert (glued["f*_var"].values > 0).all()
