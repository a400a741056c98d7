"""CPU: the batched orchestrator's host side -- tile selection (bit-exact fp64 semantics of
GPSat/dataloader.py:2352-2447 and GPSat/prediction_locations.py:18-43,208-281), the reference's result-table
layout (GPSat/local_experts.py:691-747) and the run() bookkeeping (min_obs stubs, resume, load_params) -- plus the
tile-membership / optimum known answers printed in docs/notebooks/1d_local_expert_model_part_2.ipynb."""
import numpy as np
import pandas as pd
import pytest

from gpsat_amd.engine import BatchResult
from gpsat_amd.local_experts import (BatchedLocalExpertOI, LocalSelector, PredictionLocations, get_results,
                                     max_dist_bool)
from oracle import gp_oracle as go


def _notebook_data():
    np.random.seed(0)
    N, noise_std = 100, 0.05
    X_grid = np.linspace(0.1, 0.6, 100)
    X = np.random.uniform(0.1, 0.6, (N,))
    eps = noise_std * np.random.randn(N)
    y = np.sin(1 / X) + eps
    return pd.DataFrame({"x": X, "y": y}), X_grid, noise_std


def _select(radius):
    return [{"col": "x", "comp": "<=", "val": radius}, {"col": "x", "comp": ">=", "val": -radius}]


class OracleEngine:
    """Engine stand-in for CPU tests: the fp64 oracle behind the packed-batch interface (test-only)."""
    device_name = "cpu-oracle (tests only)"
    device_id = 0

    def __init__(self):
        self.calls = []

    def fit_predict_batch(self, *, D, obs_off, X, y, pred_off, Xs, theta0, lo, hi, trainable, kernel, optimiser,
                          max_iter, **kw):
        self.calls.append(dict(T=len(obs_off) - 1, obs_off=np.array(obs_off), pred_off=np.array(pred_off)))
        o = go.fit_predict_batch(go.KERNEL_IDS[kernel], D, obs_off, X.astype(np.float64), y.astype(np.float64), pred_off,
                                 Xs.astype(np.float64), theta0, lo, hi, np.asarray(trainable, bool), max_iter=max_iter,
                                 optimise=optimiser != "none")
        T = len(obs_off) - 1
        fc = cov_off = None
        if kw.get("full_cov"):
            Pt = np.diff(pred_off)
            cov_off = np.concatenate([[0], np.cumsum(Pt * Pt)]).astype(np.int64)
            fc = np.zeros(int(cov_off[-1]))
            for t in range(T):
                a, b, pa, pb = obs_off[t], obs_off[t + 1], pred_off[t], pred_off[t + 1]
                if pb > pa:
                    fcov, _ = go.predict_cov(go.KERNEL_IDS[kernel], X[a:b].astype(np.float64), y[a:b].astype(np.float64),
                                             Xs[pa:pb].astype(np.float64), o["theta"][t])
                    fc[cov_off[t]:cov_off[t + 1]] = fcov.reshape(-1)
        return BatchResult(theta=o["theta"], nll=o["nll"], status=np.where(o["success"], 0, 1).astype(np.int32),
                           n_eval=o["n_eval"].astype(np.int32), f_mean=o["f_mean"].astype(np.float32),
                           f_var=o["f_var"].astype(np.float32), y_var=o["y_var"].astype(np.float32), f_cov=fc, cov_off=cov_off)


@pytest.mark.parametrize("radius,locs,nobs,npred,ls", [
    (0.15, [0.25, 0.45], [62, 59], [60, 60], [0.0321035488284147, 0.16317534178011256]),
    (0.10, [0.2, 0.3, 0.4, 0.5], [41, 37, 44, 38], [40, 40, 40, 40],
     [0.03354575999266631, 0.0887015158885798, 0.1793349088554155, 0.2911974656858733]),
])
def test_notebook_tile_membership_and_optima(radius, locs, nobs, npred, ls):
    """`number obs` 62/59 and 41/37/44/38, and the trained lengthscales, as printed by the reference notebook."""
    df, X_grid, noise_std = _notebook_data()
    sel = LocalSelector(df, _select(radius))
    pl = PredictionLocations(method="from_dataframe", coords_col=["x"], df=pd.DataFrame({"x": X_grid}),
                             max_dist=radius + 1e-8)
    for loc, n_expected, p_expected, ls_expected in zip(locs, nobs, npred, ls):
        m = sel.mask({"x": loc})
        assert m.sum() == n_expected
        assert len(pl(np.array([loc]))) == p_expected
        d = df.loc[m]
        np.testing.assert_array_equal(d.index.values, np.sort(d.index.values))      # source row order kept
        o = go.OracleGPR(d[["x"]].values, d[["y"]].values, kernel="RBF", noise_variance=noise_std ** 2)
        assert o.optimise_parameters(fixed_params=["likelihood_variance"])
        assert abs(o.get_parameters()["lengthscales"][0] - ls_expected) < 1e-6


def test_selection_edge_semantics():
    df = pd.DataFrame({"x": [0.0, 3.0, 0.0, 5.0, -3.0], "y": [0.0, 4.0, 5.0, 0.0, -4.0], "t": [0, 1, 2, 3, 10.0]})
    # multi-column: Euclidean ball, INCLUSIVE at r for "<" as well as "<=" (dataloader.py:2424,2439-2444)
    for comp in ("<", "<="):
        m = LocalSelector(df, [{"col": ["x", "y"], "comp": comp, "val": 5.0}]).mask({"x": 0.0, "y": 0.0, "t": 0.0})
        assert m.tolist() == [True, True, True, True, True]
    m = LocalSelector(df, [{"col": ["x", "y"], "comp": "<", "val": 4.999}]).mask({"x": 0.0, "y": 0.0, "t": 0.0})
    assert m.tolist() == [True, False, False, False, False]
    # 1-D: ref + val offsets with the literal comparison
    sel = LocalSelector(df, [{"col": "t", "comp": "<=", "val": 2}, {"col": "t", "comp": ">", "val": -2}])
    assert sel.mask({"x": 0, "y": 0, "t": 1.0}).tolist() == [True, True, True, True, False]
    assert sel.mask({"x": 0, "y": 0, "t": 2.0}).tolist() == [False, True, True, True, False]
    with pytest.raises(AssertionError):
        LocalSelector(df, [{"col": ["x", "y"], "comp": ">=", "val": 1.0}])
    with pytest.raises(AssertionError):
        LocalSelector(df, [{"col": "x", "comp": "!=", "val": 1.0}])
    # prediction locations: STRICT < on the squared distance (prediction_locations.py:37,43)
    loc = np.array([[3.0, 4.0], [3.0, 3.9], [6.0, 0.0]])
    assert max_dist_bool(loc, np.zeros(2), 5.0).tolist() == [False, True, False]
    # frame lacking a coordinate: filled from the expert location (prediction_locations.py:262-271)
    pl = PredictionLocations(method="from_dataframe", coords_col=["x", "y", "t"],
                             df=pd.DataFrame({"x": [0.0, 1.0, 9.0], "y": [0.0, 1.0, 9.0]}), max_dist=2.0)
    out = pl(np.array([0.5, 0.5, 7.0]))
    np.testing.assert_array_equal(out, [[0.0, 0.0, 7.0], [1.0, 1.0, 7.0]])
    assert PredictionLocations(method="expert_loc", coords_col=["x", "y"])(np.array([1.0, 2.0])).tolist() == [[1.0, 2.0]]
    # shift_arrays: the 'ij' mesh of the per-coordinate shifts (absent coordinates shift by 0) added to the expert
    # location (prediction_locations.py:182-206)
    sa = PredictionLocations(method="shift_arrays", coords_col=["x", "y", "t"], x=np.array([-1.0, 0.0, 1.0]), y=[0.0, 10.0])
    np.testing.assert_array_equal(sa(np.array([5.0, 6.0, 7.0])),
                                  [[4, 6, 7], [4, 16, 7], [5, 6, 7], [5, 16, 7], [6, 6, 7], [6, 16, 7]])
    # from_source: loaded once, duplicates dropped, then as from_dataframe (prediction_locations.py:83-101)
    fs = PredictionLocations(method="from_source", coords_col=["x", "y"], max_dist=2.0,
                             load_kwargs={"source": pd.DataFrame({"x": [0.0, 0.0, 1.0, 9.0], "y": [0.0, 0.0, 1.0, 9.0]})})
    np.testing.assert_array_equal(fs(np.array([0.5, 0.5])), [[0.0, 0.0], [1.0, 1.0]])
    with pytest.raises(ValueError):
        PredictionLocations(method="no_such_method", coords_col=["x"])


def _configs(df, X_grid, radius, locs, noise_std, **model_extra):
    return dict(expert_loc_config={"source": pd.DataFrame({"x": locs})},
                data_config={"data_source": df, "obs_col": ["y"], "coords_col": ["x"], "local_select": _select(radius)},
                model_config={"oi_model": "HipGPRModel",
                              "init_params": {"kernel": "RBF", "noise_variance": noise_std ** 2},
                              "optim_kwargs": {"fixed_params": ["likelihood_variance"]}, **model_extra},
                pred_loc_config={"method": "from_dataframe", "df": pd.DataFrame({"x": X_grid}),
                                 "max_dist": radius + 1e-8})


def test_run_tables_layout_resume_and_load_params(tmp_path):
    df, X_grid, noise_std = _notebook_data()
    eng = OracleEngine()
    cfg = _configs(df, X_grid, 0.1, [0.2, 0.3, 0.4, 0.5, 5.0], noise_std)
    store = str(tmp_path / "store")
    oi = BatchedLocalExpertOI(engine=eng, **cfg)
    tabs = oi.run(store_path=store, min_obs=38)
    assert len(eng.calls) == 1 and eng.calls[0]["T"] == 3                      # ONE packed call for all tiles
    assert np.diff(eng.calls[0]["obs_off"]).tolist() == [41, 44, 38]
    rd = tabs["run_details"]
    # x = 5.0 has no prediction locations: skipped, nothing stored (local_experts.py:962-965);
    # x = 0.3 has 37 < min_obs observations: stub row (local_experts.py:988-1012)
    assert rd.index.name == "x" and sorted(rd.index.tolist()) == [0.2, 0.3, 0.4, 0.5]
    assert list(rd.columns) == ["_dim_0", "num_obs", "run_time", "objective_value", "parameters_optimised",
                                "optimise_success", "model", "device", "config_id"]
    assert rd.loc[[0.2, 0.3, 0.4, 0.5], "num_obs"].tolist() == [41, 37, 44, 38]
    stub = rd.loc[0.3]
    assert np.isnan(stub["objective_value"]) and not stub["optimise_success"] and stub["device"] == ""
    assert rd.loc[[0.2, 0.4, 0.5], "optimise_success"].tolist() == [True] * 3
    pr = tabs["preds"]
    assert list(pr.columns) == ["_dim_0", "f*", "f*_var", "y_var", "f_bar", "pred_loc_x"]
    assert pr.loc[0.2]["_dim_0"].tolist() == list(range(40)) and len(pr) == 120
    assert np.allclose(pr["y_var"] - pr["f*_var"], noise_std ** 2, atol=1e-6) and (pr["f_bar"] == 0).all()
    ls = tabs["lengthscales"]
    assert list(ls.columns) == ["_dim_0", "lengthscales"] and ls.index.tolist() == [0.2, 0.4, 0.5]
    np.testing.assert_allclose(ls["lengthscales"].values,
                               [0.03354575999266631, 0.1793349088554155, 0.2911974656858733], atol=1e-6)
    assert tabs["likelihood_variance"]["likelihood_variance"].tolist() == pytest.approx([noise_std ** 2] * 3)
    # the store has the same tables + expert_locs + config; a second run resumes (nothing left to do)
    on_disk = get_results(store)
    assert {"run_details", "preds", "lengthscales", "kernel_variance", "likelihood_variance", "expert_locs"} <= set(on_disk)
    pd.testing.assert_frame_equal(on_disk["preds"], pr)
    eng2 = OracleEngine()
    tabs2 = BatchedLocalExpertOI(engine=eng2, **cfg).run(store_path=store, min_obs=38)
    assert len(eng2.calls) == 0 and len(tabs2["run_details"]) == 0
    assert len(get_results(store)["run_details"]) == 4
    # predict-only re-run with parameters loaded from the store (optimise=False): same predictions, objective
    # evaluated, parameters not saved again to a different suffix unless asked
    eng3 = OracleEngine()
    cfg3 = _configs(df, X_grid, 0.1, [0.2, 0.3, 0.4, 0.5], noise_std, load_params={"file": store, "table_suffix": ""})
    # x = 0.3 has no stored parameters -> that tile is skipped (local_experts.py:1099-1101)
    tabs3 = BatchedLocalExpertOI(engine=eng3, **cfg3).run(store_path=str(tmp_path / "store2"), optimise=False,
                                                          table_suffix="_P")
    np.testing.assert_allclose(tabs3["preds_P"]["f*"].values, pr["f*"].values, atol=2e-6)
    assert not tabs3["run_details_P"]["optimise_success"].any() and not tabs3["run_details_P"]["parameters_optimised"].any()
    np.testing.assert_allclose(tabs3["lengthscales_P"]["lengthscales"].values, ls["lengthscales"].values)


def test_unsupported_configs_fail_loudly():
    df, X_grid, noise_std = _notebook_data()
    base = _configs(df, X_grid, 0.1, [0.2], noise_std)
    # load_params.previous together with a file: the reference reads the file and never uses the running average
    # (`if file is not None ... elif previous is not None`, local_experts.py:553-609)
    both = dict(base)
    both["model_config"] = {**base["model_config"], "load_params": {"previous": True, "file": "somewhere", "table_suffix": "_S"}}
    oi = BatchedLocalExpertOI(engine=OracleEngine(), **both)
    assert oi.use_previous is False and oi.load_params == {"file": "somewhere", "table_suffix": "_S"} and "previous" in oi._lp_keys
    # ... and direct values beside `previous` are never applied
    oi = BatchedLocalExpertOI(engine=OracleEngine(), **{**base, "model_config": {**base["model_config"],
                                                        "load_params": {"previous": False, "lengthscales": [0.3]}}})
    assert oi.use_previous is False and oi.load_params is None
    bad = dict(base)
    bad["model_config"] = {**base["model_config"], "oi_model": "GPflowSVGPModel"}
    with pytest.raises(NotImplementedError):
        BatchedLocalExpertOI(engine=OracleEngine(), **bad)
    bad = dict(base)
    bad["model_config"] = {**base["model_config"], "replacement_threshold": 10, "replacement_model": "GPflowSGPRModel"}
    with pytest.raises(NotImplementedError):
        BatchedLocalExpertOI(engine=OracleEngine(), **bad)


def test_replacement_model_for_small_tiles(tmp_path):
    """Tiles with fewer than `replacement_threshold` observations run with the replacement settings
    (GPSat/local_experts.py:339-346,1021-1041): here another covariance function and a fixed lengthscale; one engine
    call per profile, tables in expert order."""
    df, X_grid, noise_std = _notebook_data()
    cfg = _configs(df, X_grid, 0.1, [0.2, 0.3, 0.4, 0.5], noise_std)          # 41 / 37 / 44 / 38 observations
    cfg["model_config"] = {**cfg["model_config"], "replacement_threshold": 40,
                           "replacement_init_params": {"kernel": "Matern32", "noise_variance": noise_std ** 2,
                                                       "kernel_kwargs": {"lengthscales": [0.07]}},
                           "replacement_optim_kwargs": {"fixed_params": ["lengthscales", "likelihood_variance"]}}
    eng = OracleEngine()
    oi = BatchedLocalExpertOI(engine=eng, **cfg)
    tabs = oi.run(store_path=str(tmp_path / "s"))
    assert [c["T"] for c in eng.calls] == [2, 2]                                 # main profile, replacement profile
    rd = tabs["run_details"]
    assert rd.index.tolist() == [0.2, 0.3, 0.4, 0.5] and rd["num_obs"].tolist() == [41, 37, 44, 38]
    ls = tabs["lengthscales"]["lengthscales"]
    assert ls.loc[0.3] == pytest.approx(0.07) and ls.loc[0.5] == pytest.approx(0.07)      # fixed in the replacement
    assert abs(ls.loc[0.2] - 0.03354575999266631) < 1e-5 and abs(ls.loc[0.4] - 0.1793349088554155) < 1e-5
    # the replacement tiles really used Matern-3/2: their objective equals the oracle's Matern-3/2 objective
    m = (df["x"] <= 0.3 + 0.1) & (df["x"] >= 0.3 - 0.1)
    d = df.loc[m]
    o = go.OracleGPR(d[["x"]].values, d[["y"]].values, kernel="Matern32", noise_variance=noise_std ** 2)
    o.set_parameters(lengthscales=[0.07], kernel_variance=float(tabs["kernel_variance"]["kernel_variance"].loc[0.3]))
    assert o.get_objective_function_value() == pytest.approx(rd["objective_value"].loc[0.3], abs=1e-4)   # fp32-packed inputs


def test_ragged_rows_reads_like_a_list_of_arrays():
    """`RaggedRows` (the prediction coordinates of all experts in one array): indexing, iteration and `take` of consecutive,
    scattered, repeated and empty selections equal the list-of-arrays form they replaced."""
    from gpsat_amd.local_experts import RaggedRows
    rng = np.random.default_rng(0)
    counts = np.array([3, 0, 5, 1, 0, 0, 4, 2])
    rows = [rng.normal(size=(c, 3)) for c in counts]
    off = np.concatenate([[0], np.cumsum(counts)])
    rr = RaggedRows(np.concatenate(rows), off)
    assert len(rr) == len(rows) and (rr.counts == counts).all()
    for a, b in zip(rr, rows):
        assert np.array_equal(a, b)
    for items in ([0, 1, 2, 3], [2], [7, 0, 2], [1, 4, 5], [], [3, 3, 6], list(range(8))):
        want = np.concatenate([rows[i] for i in items]) if len(items) else np.zeros((0, 3))
        got = rr.take(np.asarray(items, dtype=np.int64))
        assert got.shape == want.shape and np.array_equal(got, want), items
