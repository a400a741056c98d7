"""GPU: the orchestrator and the sharded path on the device.

* BASELINE.json configs[3] on one GPU: 100 000 synthetic N=500 tiles (RBF, D=3) as ONE global tile list, split into two
  logical shards by the LPT partition, each shard packed and run through the C ABI, results merged into the reference's
  tile order -- bit-identical to the unsharded run, spot tiles checked against the oracle
  (the reference fixes the expert order at /root/reference/GPSat/local_experts.py:416-420 and loops over it at :930);
* a 4096-expert orchestrator run killed mid-way resumes from the committed waves and ends with bit-identical tables
  (the reference's store_every / resume contract, local_experts.py:500-548,908-912);
* coordinates far from the origin (t ~ 1e4 length scales away): the per-tile centring before the fp32 cast keeps the
  stated fp64 -> fp32 tolerance;
* dtype="f64" through the orchestrator (predict-only with loaded parameters, BASELINE configs[4] in small).
"""
import os

import numpy as np
import pandas as pd
import pytest

from gpsat_amd import sharding
from gpsat_amd import synthetic as syn
from oracle import gp_oracle as go

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from gpsat_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_configs3_100k_tiles_two_logical_shards(eng):
    T, N, P, D, kid, NPROTO = 100_000, 500, 20, 3, 0, 500
    proto = [syn.make_tile(31_000 + j, N, P, D, kid) for j in range(NPROTO)]
    # every tile is distinct: prototype j = t % NPROTO with the observations scaled by a per-replica factor, so a tile
    # returned in the wrong slot cannot go unnoticed
    scale = 1.0 + 0.01 * (np.arange(T) // NPROTO)
    X = np.concatenate([p[0] for p in proto]).astype(np.float32)
    X = np.tile(X, (T // NPROTO, 1))
    Xs = np.tile(np.concatenate([p[2] for p in proto]).astype(np.float32), (T // NPROTO, 1))
    y = (np.tile(np.concatenate([p[1] for p in proto]), T // NPROTO) * np.repeat(scale, N)).astype(np.float32)
    lo, hi = syn.default_bounds(T, D)
    batch = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
                 X=X, y=y, Xs=Xs, theta0=np.ones((T, D + 2)), lo=lo, hi=hi)
    kw = dict(kernel="RBF", optimiser="lbfgs", max_iter=20)
    whole = eng.fit_predict_batch(D=D, obs_off=batch["obs_off"], X=X, y=y, pred_off=batch["pred_off"], Xs=Xs,
                                  theta0=batch["theta0"], lo=lo, hi=hi, **kw)
    fixed_g, preds_g, pred_off_g, shard_res = sharding.run_sharded(eng, batch, world_size=2, rank=None, **kw)
    parts = sharding.partition_tiles(np.full(T, N), np.full(T, P), 2)
    assert abs(len(parts[0]) - len(parts[1])) <= 1 and len(shard_res) == 2
    # (a) order and bit-identity against the unsharded run
    H = D + 2
    np.testing.assert_array_equal(fixed_g[:, :H], whole.theta)
    np.testing.assert_array_equal(fixed_g[:, H], whole.nll)
    np.testing.assert_array_equal(fixed_g[:, H + 1], whole.status)
    np.testing.assert_array_equal(fixed_g[:, H + 2], whole.n_eval)
    np.testing.assert_array_equal(pred_off_g, batch["pred_off"])
    np.testing.assert_array_equal(preds_g[:, 0], whole.f_mean)
    np.testing.assert_array_equal(preds_g[:, 1], whole.f_var)
    np.testing.assert_array_equal(preds_g[:, 2], whole.y_var)
    assert np.isin(whole.status, (0, 1, 6)).all() and (whole.status <= 1).mean() > 0.9
    # replicas differ (the per-replica scaling reached the kernels)
    assert len(np.unique(np.round(whole.nll[::NPROTO][:50], 6))) > 40
    # (b) spot tiles against the oracle at the returned parameters
    for t in [0, 1, 499, 500, 31_337, 50_000, 77_777, 99_999]:
        Xo, yo = X[t * N:(t + 1) * N].astype(np.float64), y[t * N:(t + 1) * N].astype(np.float64)
        th = fixed_g[t, :H]
        nll, _ = go.nll_and_grad(kid, Xo, yo, th, want_grad=False)
        f, fv, yv = go.predict(kid, Xo, yo, Xs[t * P:(t + 1) * P].astype(np.float64), th)
        assert abs(nll - fixed_g[t, H]) <= 2e-5 * N + 2e-6 * abs(nll)
        sl = slice(t * P, (t + 1) * P)
        assert np.max(np.abs(preds_g[sl, 0] - f)) <= 2e-3 * np.abs(yo).max()
        assert np.max(np.abs(preds_g[sl, 1] - fv)) <= 2e-3 * th[D] + 1e-6


def _grid_problem(n_side=64, m=120_000, seed=3):
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0, 1, (m, 2))
    t = rng.uniform(18_000, 18_008, m)                            # days since 1970: far from the origin
    z = np.sin(6 * xy[:, 0]) * np.cos(5 * xy[:, 1]) + 0.05 * (t - 18_004) + 0.1 * rng.standard_normal(m)
    df = pd.DataFrame({"x": xy[:, 0], "y": xy[:, 1], "t": t, "z": z})
    g = (np.arange(n_side) + 0.5) / n_side
    xl = pd.DataFrame([(a, b, 18_004.0) for a in g for b in g], columns=["x", "y", "t"])
    return dict(
        expert_loc_config={"source": xl},
        data_config={"data_source": df, "obs_col": "z", "coords_col": ["x", "y", "t"],
                     "local_select": [{"col": ["x", "y"], "comp": "<", "val": 0.02},
                                      {"col": "t", "comp": "<=", "val": 4}, {"col": "t", "comp": ">=", "val": -4}]},
        model_config={"oi_model": "HipGPRModel",
                      "init_params": {"kernel": "Matern32", "obs_mean": "local", "coords_scale": [0.01, 0.01, 1.0]},
                      "constraints": {"lengthscales": {"low": [1e-8, 1e-8, 1e-8], "high": [0.12, 0.12, 9.0]}},
                      "optim_kwargs": {"max_iter": 30}},
        pred_loc_config={"method": "expert_loc"})


class _Dying:
    """wraps the engine: the k-th fit_predict_batch call raises (a fault in the middle of the sweep)"""

    def __init__(self, eng, die_at):
        self._eng, self._n, self._die = eng, 0, die_at

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def fit_predict_batch(self, **kw):
        self._n += 1
        if self._n == self._die:
            raise RuntimeError("simulated fault")
        return self._eng.fit_predict_batch(**kw)


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        da = a[k].drop(columns=[c for c in ("run_time", "config_id") if c in a[k].columns])
        db = b[k].drop(columns=[c for c in ("run_time", "config_id") if c in b[k].columns])
        pd.testing.assert_frame_equal(da, db, check_exact=True)


def test_engine_pool_and_chunking_do_not_change_a_bit(eng, tmp_path):
    """Two engines (HIP streams) taking the chunks of a wave in turn, four packing threads, chunks of 300 tiles -- launches
    below and above the CU count, i.e. both builds of the fp32 kernels -- against one engine, one thread, one call per wave:
    the same tables, bit for bit (run_time aside)."""
    from gpsat_amd.local_experts import BatchedLocalExpertOI
    cfg = _grid_problem(n_side=40, m=60_000, seed=5)
    a = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg)
    a.engine_workers, a.pack_threads = 1, 1
    ta = a.run(store_path=str(tmp_path / "a"), store_every=4096)
    b = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg)
    assert b.engine_workers == 2 and b.pack_threads == 4
    tb = b.run(store_path=str(tmp_path / "b"), store_every=700, engine_chunk=300)
    assert len(b._extra_engines) == 1 and len(ta["run_details"]) == 1600
    _same(ta, tb)


def test_kill_and_resume_4096_experts(eng, tmp_path):
    from gpsat_amd.local_experts import BatchedLocalExpertOI, get_results
    cfg = _grid_problem()
    oi = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg)
    full = oi.run(store_path=str(tmp_path / "full"), store_every=512)
    assert len(full["run_details"]) == 4096 and full["run_details"]["num_obs"].between(60, 260).all()
    assert full["run_details"]["optimise_success"].mean() > 0.8
    print("orchestrator 4096 experts:", {k: round(v, 3) for k, v in oi.timings.items() if k != "calls"}, "total s", round(oi.run_seconds, 3))
    store = str(tmp_path / "killed")
    with pytest.raises(RuntimeError):
        BatchedLocalExpertOI(engine=_Dying(eng, 4), device_select=True, **cfg).run(store_path=store, store_every=512)
    assert len(get_results(store)["run_details"]) == 3 * 512                # three committed waves survive
    rest = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg).run(store_path=store, store_every=512)
    assert len(rest["run_details"]) == 4096 - 3 * 512
    resumed = get_results(store)
    _same(full, {k: v for k, v in resumed.items() if k in full})
    # the same sweep as two logical shards written into one store, read back in expert order
    store2 = str(tmp_path / "sharded")
    merged = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg).run(store_path=store2, store_every=512, world_size=2)
    _same(full, merged)
    assert any(".r001." in f for f in os.listdir(store2))                    # both shards committed their own parts
    _same(full, {k: v for k, v in get_results(store2, expert_order=True).items() if k in full})


def test_far_from_origin_coordinates_fp32(eng):
    """t ~ 18 000 with length scales of a few units: X is handed over in fp64 and centred per tile before the fp32 cast
    (engine.centre_tiles); the fp32 results keep the stated tolerance against the fp64 oracle on the ORIGINAL coordinates."""
    T, N, P, D, kid = 3, 300, 24, 3, 2
    b = syn.make_batch(T, N, P, D, kid, base_seed=811, dtype=np.float64)
    off = np.array([4.0e3, -2.5e3, 1.8e4])
    X, Xs = b["X"] + off, b["Xs"] + off
    th = b["truth"].copy()
    r = eng.fit_predict_batch(D=D, obs_off=b["obs_off"], X=X, y=b["y"], pred_off=b["pred_off"], Xs=Xs, theta0=th,
                              kernel="Matern32", optimiser="none", want_grad=True)
    for t in range(T):
        a, e = b["obs_off"][t], b["obs_off"][t + 1]
        pa, pe = b["pred_off"][t], b["pred_off"][t + 1]
        nll, g = go.nll_and_grad(kid, X[a:e], b["y"][a:e], th[t])
        f, fv, _ = go.predict(kid, X[a:e], b["y"][a:e], Xs[pa:pe], th[t])
        assert abs(r.nll[t] - nll) <= 2e-5 * N + 2e-6 * abs(nll)
        np.testing.assert_array_less(np.abs(r.grad[t] - g), 2e-3 * (np.abs(g) + np.abs(g).max()) + 1e-9)
        assert np.max(np.abs(r.f_mean[pa:pe] - f)) <= 2e-3 * np.abs(b["y"][a:e]).max()
        assert np.max(np.abs(r.f_var[pa:pe] - fv)) <= 2e-3 * th[t, D] + 1e-6


def test_orchestrator_fp64_predict_only_with_loaded_parameters(eng, tmp_path):
    """BASELINE configs[4] in small, through the orchestrator: fp64 kernels, optimise=False, per-tile parameters loaded
    from a store; objective and predictions equal the fp64 oracle's to 1e-8."""
    from gpsat_amd.local_experts import BatchedLocalExpertOI, ResultStore
    cfg = _grid_problem(n_side=4, m=20_000)
    cfg["data_config"]["local_select"][0]["val"] = 0.08
    cfg["pred_loc_config"] = {"method": "shift_arrays", "x": np.array([-0.01, 0.0, 0.01]), "y": np.array([0.0, 0.01])}
    xl = cfg["expert_loc_config"]["source"]
    rng = np.random.default_rng(0)
    cc = ["x", "y", "t"]
    ls = pd.concat([xl.assign(_dim_0=d, lengthscales=rng.uniform(2.0, 6.0, len(xl))) for d in range(3)]).set_index(cc)
    kv = xl.assign(_dim_0=0, kernel_variance=rng.uniform(0.5, 1.5, len(xl))).set_index(cc)
    lv = xl.assign(_dim_0=0, likelihood_variance=rng.uniform(0.01, 0.05, len(xl))).set_index(cc)
    src = str(tmp_path / "params")
    st = ResultStore(src)
    st.put("lengthscales_SMOOTHED", ls), st.put("kernel_variance_SMOOTHED", kv), st.put("likelihood_variance_SMOOTHED", lv)
    cfg["model_config"]["load_params"] = {"file": src, "table_suffix": "_SMOOTHED"}
    oi = BatchedLocalExpertOI(engine=eng, dtype="f64", **cfg)
    tabs = oi.run(optimise=False)
    rd, pr = tabs["run_details"], tabs["preds"]
    assert len(rd) == 16 and not rd["optimise_success"].any()
    df = cfg["data_config"]["data_source"]
    scale = np.array([0.01, 0.01, 1.0])
    for i in (0, 5, 15):
        loc = xl.iloc[i]
        m = ((df["x"] - loc["x"]) ** 2 + (df["y"] - loc["y"]) ** 2 <= 0.08 ** 2) & (np.abs(df["t"] - loc["t"]) <= 4)
        d = df[m]
        key = tuple(loc[cc])
        th = np.concatenate([ls.loc[[key]].sort_values("_dim_0")["lengthscales"].values,
                             kv.loc[[key]]["kernel_variance"].values, lv.loc[[key]]["likelihood_variance"].values])
        Xo, yo = d[cc].values / scale, d["z"].values - d["z"].values.mean()
        nll, _ = go.nll_and_grad(2, Xo, yo, th, want_grad=False)
        assert rd.loc[[key]]["num_obs"].values[0] == len(d)
        assert rd.loc[[key]]["objective_value"].values[0] == pytest.approx(nll, rel=1e-9, abs=1e-8)
        p = pr.loc[[key]]
        Xp = p[[f"pred_loc_{c}" for c in cc]].values / scale
        f, fv, yv = go.predict(2, Xo, yo, Xp, th)
        np.testing.assert_allclose(p["f*"].values, f, rtol=0, atol=1e-8)
        np.testing.assert_allclose(p["f*_var"].values, fv, rtol=0, atol=1e-9)
        np.testing.assert_allclose(p["f_bar"].values, d["z"].values.mean(), rtol=1e-15)


def test_orchestrator_four_input_dimensions_fp64(eng, tmp_path):
    """Four coordinate columns (x, y, t and a depth-like d) through the orchestrator in fp64: tile membership by the
    reference's predicate, and at the parameters each tile's optimiser returned the objective and the predictions equal
    the oracle's (the D = 4 kernels; the reference takes any number of coordinate columns, GPSat/models/base_model.py:134-189)."""
    from gpsat_amd.local_experts import BatchedLocalExpertOI
    rng = np.random.default_rng(4)
    M = 6000
    df = pd.DataFrame({"x": rng.uniform(0, 1, M), "y": rng.uniform(0, 1, M), "t": rng.uniform(0, 6, M), "d": rng.uniform(0, 2, M)})
    df["z"] = (np.sin(6 * df["x"]) * np.cos(5 * df["y"]) + 0.3 * np.sin(df["t"]) + 0.2 * df["d"] ** 2 + 0.05 * rng.normal(size=M))
    xl = pd.DataFrame({"x": [0.3, 0.7, 0.5], "y": [0.4, 0.6, 0.5], "t": [3.0, 2.0, 4.0], "d": [1.0, 0.5, 1.5]})
    cc = ["x", "y", "t", "d"]
    oi = BatchedLocalExpertOI(
        expert_loc_config={"source": xl},
        data_config={"data_source": df, "obs_col": "z", "coords_col": cc,
                     "local_select": [{"col": ["x", "y"], "comp": "<=", "val": 0.2}, {"col": "t", "comp": "<=", "val": 2.0},
                                      {"col": "t", "comp": ">=", "val": -2.0}]},
        model_config={"oi_model": "HipGPRModel", "init_params": {"kernel": "Matern32", "obs_mean": "local"},
                      "constraints": {"lengthscales": {"low": [1e-3] * 4, "high": [5.0] * 4}}},
        pred_loc_config={"method": "expert_loc"}, engine=eng, dtype="f64")
    tabs = oi.run(store_path=str(tmp_path / "s4"))
    rd, pr = tabs["run_details"], tabs["preds"]
    assert len(rd) == 3 and rd["optimise_success"].all()
    ls = tabs["lengthscales"]
    assert sorted(ls["_dim_0"].unique().tolist()) == [0, 1, 2, 3]
    for i in range(3):
        loc = xl.iloc[i]
        m = ((df["x"] - loc["x"]) ** 2 + (df["y"] - loc["y"]) ** 2 <= 0.2 ** 2) & (np.abs(df["t"] - loc["t"]) <= 2.0)
        d = df[m]
        key = tuple(loc[cc])
        assert rd.loc[[key]]["num_obs"].values[0] == len(d)
        th = np.concatenate([ls.loc[[key]].sort_values("_dim_0")["lengthscales"].values,
                             tabs["kernel_variance"].loc[[key]]["kernel_variance"].values,
                             tabs["likelihood_variance"].loc[[key]]["likelihood_variance"].values])
        Xo, yo = d[cc].values, d["z"].values - d["z"].values.mean()
        nll, _ = go.nll_and_grad(2, Xo, yo, th, want_grad=False)
        assert rd.loc[[key]]["objective_value"].values[0] == pytest.approx(nll, rel=1e-9, abs=1e-7)
        p = pr.loc[[key]]
        f, fv, _ = go.predict(2, Xo, yo, p[[f"pred_loc_{c}" for c in cc]].values, th)
        np.testing.assert_allclose(p["f*"].values, f, rtol=0, atol=1e-7)
        np.testing.assert_allclose(p["f*_var"].values, fv, rtol=0, atol=1e-8)
