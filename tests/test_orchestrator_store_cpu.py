"""CPU: the orchestrator's store / resume / sharding contract.

* append-only, atomically committed parts flushed every ``store_every`` expert locations (the reference appends to its
  HDFStore every ``store_every`` tiles, GPSat/local_experts.py:500-548,1252-1257);
* a run killed mid-way resumes after the last committed wave and ends with the same tables as an uninterrupted run
  (GPSat/local_experts.py:475-497,908-912);
* a world_size-2 tile-sharded run (gloo) gathers to rank 0 the same tables, in the reference's expert order, as an
  unsharded run (GPSat/local_experts.py:416-420 fixes that order);
* ``load_params`` variants of GPSat/local_experts.py:553-604 (``index_adjust``, parameters given directly);
* ``check_prev_oi_config`` (GPSat/utils.py:1276-1327).
The engine is the oracle behind the packed-batch interface (tests only)."""
import os
import socket

import numpy as np
import pandas as pd
import pytest

from gpsat_amd import sharding
from gpsat_amd.local_experts import BatchedLocalExpertOI, ResultStore, check_prev_oi_config, get_results
from test_local_experts_cpu import OracleEngine, _configs, _notebook_data


def _grid_case(n_locs=9):
    """2-D observations on [0,1]^2 x one time coordinate, experts on a small grid."""
    rng = np.random.default_rng(5)
    M = 900
    df = pd.DataFrame({"x": rng.uniform(0, 1, M), "y": rng.uniform(0, 1, M), "t": rng.uniform(-1, 1, M)})
    df["z"] = np.sin(4 * df["x"]) * np.cos(3 * df["y"]) + 0.1 * df["t"] + 0.05 * rng.standard_normal(M)
    g = np.linspace(0.2, 0.8, int(np.sqrt(n_locs)))
    xl = pd.DataFrame([(a, b, 0.0) for a in g for b in g], columns=["x", "y", "t"])
    return dict(
        expert_loc_config={"source": xl},
        data_config={"data_source": df, "obs_col": "z", "coords_col": ["x", "y", "t"],
                     "local_select": [{"col": ["x", "y"], "comp": "<", "val": 0.2}, {"col": "t", "comp": "<=", "val": 1},
                                      {"col": "t", "comp": ">=", "val": -1}]},
        model_config={"oi_model": "HipGPRModel", "init_params": {"kernel": "Matern32", "obs_mean": "local",
                                                                 "coords_scale": [0.5, 0.5, 2.0]},
                      "constraints": {"lengthscales": {"low": [1e-8, 1e-8, 1e-8], "high": [2.0, 2.0, 4.0]}},
                      "optim_kwargs": {"max_iter": 15}},
        pred_loc_config={"method": "shift_arrays", "x": np.array([-0.05, 0.0, 0.05]), "y": np.array([0.0, 0.05])})


class DyingEngine(OracleEngine):
    def __init__(self, die_at_call):
        super().__init__()
        self.die_at_call = die_at_call

    def fit_predict_batch(self, **kw):
        if len(self.calls) + 1 == self.die_at_call:
            raise RuntimeError("simulated fault in the middle of the sweep")
        return super().fit_predict_batch(**kw)


def _assert_same_tables(a, b, ignore=("run_time", "config_id")):
    assert set(a) == set(b)
    for k in a:
        da = a[k].drop(columns=[c for c in ignore if c in a[k].columns])
        db = b[k].drop(columns=[c for c in ignore if c in b[k].columns])
        pd.testing.assert_frame_equal(da, db, check_exact=True)


def test_flush_every_wave_kill_and_resume(tmp_path):
    cfg = _grid_case(9)
    full = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(store_path=str(tmp_path / "full"), store_every=2)
    on_disk = get_results(str(tmp_path / "full"))
    parts = [f for f in os.listdir(tmp_path / "full") if ".w0" in f]
    assert len([f for f in parts if f.startswith("run_details.")]) == 5           # 9 experts in waves of 2 -> 5 parts
    _assert_same_tables(full, {k: v for k, v in on_disk.items() if k in full}, ignore=())

    store = str(tmp_path / "killed")
    dying = DyingEngine(die_at_call=3)
    with pytest.raises(RuntimeError):
        BatchedLocalExpertOI(engine=dying, **cfg).run(store_path=store, store_every=2)
    partial = get_results(store)
    assert len(partial["run_details"]) == 4                                       # two committed waves survive the fault
    # an orphan part (written, never committed) is ignored by readers and removed by the next run
    pd.DataFrame({"a": [1]}).to_pickle(os.path.join(store, "preds.w000009.r000.pkl"))
    assert len(get_results(store)["preds"]) == len(partial["preds"])
    eng = OracleEngine()
    rest = BatchedLocalExpertOI(engine=eng, **cfg).run(store_path=store, store_every=2)
    assert not os.path.exists(os.path.join(store, "preds.w000009.r000.pkl"))
    assert len(rest["run_details"]) == 5 and sum(c["T"] for c in eng.calls) == 5   # only the unfinished experts ran
    resumed = get_results(store)
    _assert_same_tables(full, {k: v for k, v in resumed.items() if k in full})
    # nothing is rewritten by an append: the first two waves' files are the killed run's own
    assert ResultStore(store).read("run_details").index.equals(full["run_details"].index)


def test_load_params_direct_index_adjust_and_config_check(tmp_path):
    df, X_grid, noise_std = _notebook_data()
    cfg = _configs(df, X_grid, 0.1, [0.2, 0.4, 0.5], noise_std)
    store = str(tmp_path / "s")
    base = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(store_path=store)
    # parameters given directly: set on every tile, nothing optimised
    cfg_d = _configs(df, X_grid, 0.1, [0.2, 0.4], noise_std, load_params={"lengthscales": [0.05], "kernel_variance": 0.7})
    t = BatchedLocalExpertOI(engine=OracleEngine(), **cfg_d).run(optimise=False)
    assert t["lengthscales"]["lengthscales"].tolist() == [0.05, 0.05] and t["kernel_variance"]["kernel_variance"].tolist() == [0.7, 0.7]
    # index_adjust: the tile at x looks its parameters up at x + 0.1 (the stored 0.5 serves the tile at 0.4; 0.3 has
    # none -> the tile at 0.2 is skipped, local_experts.py:1099-1101)
    cfg_a = _configs(df, X_grid, 0.1, [0.2, 0.4], noise_std,
                     load_params={"file": store, "index_adjust": {"x": {"func": "lambda x: x + 0.1"}}})
    t = BatchedLocalExpertOI(engine=OracleEngine(), **cfg_a).run(optimise=False, table_suffix="_A")
    assert t["run_details_A"].index.tolist() == [0.4]
    assert t["lengthscales_A"]["lengthscales"].iloc[0] == base["lengthscales"]["lengthscales"].loc[0.5]
    # config bookkeeping: a second run into the same tables with another model config is refused unless told otherwise
    other = _configs(df, X_grid, 0.1, [0.2, 0.4, 0.5], noise_std)
    other["model_config"]["init_params"] = {"kernel": "Matern52", "noise_variance": noise_std ** 2}
    with pytest.raises(AssertionError):
        BatchedLocalExpertOI(engine=OracleEngine(), **other).run(store_path=store)
    BatchedLocalExpertOI(engine=OracleEngine(), **other).run(store_path=store, skip_valid_checks_on=["model"])
    check_prev_oi_config({"a": 1, "b": 2}, {"a": 1, "b": 3}, skip_valid_checks_on=["b"])
    with pytest.raises(AssertionError):
        check_prev_oi_config({"a": 1, "b": 2}, {"a": 1, "b": 3})


def test_assemble_global_logical_shards():
    rng = np.random.default_rng(1)
    T = 37
    N = rng.choice([64, 128, 500], size=T)
    P = rng.integers(0, 6, size=T)
    parts = sharding.partition_tiles(N, P, 3)
    assert sorted(np.concatenate(parts).tolist()) == list(range(T))
    shards = []
    for ids in parts:
        fx = np.stack([ids * 1.0, ids * 2.0], axis=1)
        pr = np.concatenate([np.full((P[t], 3), float(t)) + np.arange(P[t])[:, None] * 0.01 for t in ids] + [np.zeros((0, 3))])
        shards.append((fx, pr, P[ids], ids))
    fg, pg, off = sharding.assemble_global(shards, T)
    np.testing.assert_array_equal(fg[:, 0], np.arange(T))
    np.testing.assert_array_equal(off, np.concatenate([[0], np.cumsum(P)]))
    for t in range(T):
        np.testing.assert_allclose(pg[off[t]:off[t + 1], 0], t + np.arange(P[t]) * 0.01)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sharded_worker(rank, world, port, store, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _grid_case(16)
        eng = OracleEngine()
        tabs = BatchedLocalExpertOI(engine=eng, **cfg).run(store_path=store, store_every=3)    # rank / world from the group
        ret[f"tiles{rank}"] = sum(c["T"] for c in eng.calls)
        if rank == 0:
            ret["tables"] = tabs
    finally:
        dist.destroy_process_group()


def test_sharded_run_world2_matches_unsharded(tmp_path):
    import torch.multiprocessing as mp
    cfg = _grid_case(16)
    ref = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run()
    mgr = mp.Manager()
    ret = mgr.dict()
    store = str(tmp_path / "sharded")
    mp.spawn(_sharded_worker, args=(2, _free_port(), store, ret), nprocs=2, join=True)
    assert ret["tiles0"] > 0 and ret["tiles1"] > 0 and ret["tiles0"] + ret["tiles1"] == len(ref["run_details"])
    _assert_same_tables(ref, ret["tables"])                     # rank 0: global tables, reference (expert) order
    # the store holds both ranks' parts; read back in expert order it is the same tables again
    disk = get_results(store, expert_order=True)
    _assert_same_tables(ref, {k: v for k, v in disk.items() if k in ref})


def test_logical_shards_in_one_process(tmp_path):
    """world_size=3 without a process group: the three logical shards run one after the other and are merged by the
    routine that closes the gather -- the same tables as the unsharded run, each shard's parts committed under its rank."""
    cfg = _grid_case(16)
    ref = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run()
    store = str(tmp_path / "logical")
    eng = OracleEngine()
    got = BatchedLocalExpertOI(engine=eng, **cfg).run(store_path=store, store_every=4, world_size=3)
    _assert_same_tables(ref, got)
    assert {f.split(".")[2] for f in os.listdir(store) if f.startswith("run_details.w")} == {"r000", "r001", "r002"}
    _assert_same_tables(ref, {k: v for k, v in get_results(store, expert_order=True).items() if k in ref})


def test_store_is_parquet_and_reads_older_pickle_stores(tmp_path):
    """Default container: Apache Parquet parts (any parquet reader, no pandas-version binding, nothing executed on load);
    a store written as pandas pickles stays readable; an unfinished temporary is never taken for a table part."""
    import pyarrow.parquet as pq
    from gpsat_amd.local_experts import ResultStore, export_parquet
    cfg = _grid_case(9)
    store = str(tmp_path / "pq")
    full = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(store_path=store, store_every=4)
    files = os.listdir(store)
    assert any(f.startswith("preds.w000001.r000.") and f.endswith(".parquet") for f in files)
    assert not any(f.endswith(".pkl") for f in files)
    first = sorted(f for f in files if f.startswith("preds.w000001.r000."))[0]      # the wave's first row piece (one per engine call)
    t = pq.read_table(os.path.join(store, first))                                  # readable without this package
    assert {"f*", "f*_var", "y_var", "f_bar", "_dim_0", "pred_loc_x"} <= set(t.column_names)
    _assert_same_tables(full, {k: v for k, v in get_results(store).items() if k in full}, ignore=())
    # the same run into a pickle store: identical tables
    st2 = str(tmp_path / "pk")
    os.makedirs(st2)
    rs = ResultStore(st2, fmt="pickle")
    for k, v in full.items():
        rs.append(k, v)
    _assert_same_tables(full, {k: v for k, v in get_results(st2).items() if k in full}, ignore=())
    # a truncated temporary left by a crash: ignored by readers even when a marker for its wave exists,
    # removed by the next drop_uncommitted
    tmpf = os.path.join(store, ".tmp.r0.999999999.preds.w000001.r000.parquet")
    other = os.path.join(store, ".tmp.r1.999999999.preds.w000001.r001.parquet")      # ANOTHER rank's temporary, maybe in flight
    old = os.path.join(store, ".tmp.999999999.preds.w000001.r000.parquet")           # an older version's, long dead
    for f_ in (tmpf, other, old):
        open(f_, "wb").write(b"trunc")
    os.utime(old, (1.0e9, 1.0e9))
    assert ".tmp.r0.999999999.preds" not in get_results(store) and set(ResultStore(store).table_names()) >= set(full)
    ResultStore(store).drop_uncommitted()
    assert not os.path.exists(tmpf) and not os.path.exists(old)
    assert os.path.exists(other), "a rank must not remove another rank's temporaries (their PIDs may be invisible here)"
    os.remove(other)
    out = export_parquet(store, str(tmp_path / "export"))
    got = {os.path.basename(f)[:-8]: pd.read_parquet(f) for f in out}
    _assert_same_tables(full, {k: v for k, v in got.items() if k in full}, ignore=())
    assert os.path.exists(tmp_path / "export" / "oi_config.json")


def test_full_cov_tables(tmp_path):
    """pred_kwargs.full_cov=True: table preds_2 with _dim_0, _dim_1, f*_cov, y_cov in the layout
    dict_of_array_to_table(concat=True, table="preds") gives 2-D arrays (GPSat/local_experts.py:691-747,
    GPSat/models/gpflow_models.py:245-263); values against the oracle's predict_cov."""
    cfg = _grid_case(4)
    cfg["model_config"]["pred_kwargs"] = {"full_cov": True}
    eng = OracleEngine()
    tabs = BatchedLocalExpertOI(engine=eng, **cfg).run(store_path=str(tmp_path / "cov"), store_every=3)
    p1, p2 = tabs["preds"], tabs["preds_2"]
    assert list(p2.columns) == ["_dim_0", "_dim_1", "f*_cov", "y_cov"] and p2.index.names == ["x", "y", "t"]
    P = 6                                                                       # 3 x 2 shifted prediction locations
    assert len(p1) == 4 * P and len(p2) == 4 * P * P
    for loc, grp in p2.groupby(level=[0, 1, 2], sort=False):
        assert grp["_dim_0"].tolist() == np.repeat(np.arange(P), P).tolist()
        assert grp["_dim_1"].tolist() == np.tile(np.arange(P), P).tolist()
        fc = grp["f*_cov"].values.reshape(P, P)
        yc = grp["y_cov"].values.reshape(P, P)
        one = p1.loc[loc]
        np.testing.assert_allclose(np.diag(fc), one["f*_var"].values, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(np.diag(yc), one["y_var"].values, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(yc - np.diag(np.diag(yc)), fc - np.diag(np.diag(fc)), rtol=0, atol=0)
        np.testing.assert_allclose(fc, fc.T, rtol=1e-12, atol=1e-14)
    disk = get_results(str(tmp_path / "cov"), expert_order=True)
    _assert_same_tables(tabs, {k: v for k, v in disk.items() if k in tabs})
    # logical shards return the same preds_2
    got = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(world_size=2)
    _assert_same_tables(tabs, got)


def test_load_params_previous_is_the_serial_recurrence_at_chunk_1():
    """load_params.previous=True (GPSat/local_experts.py:1059-1064,1200-1217): every tile starts from the running average
    (rho 0.95) of the optima of the successfully optimised tiles before it.  engine_chunk=1 is the reference's serial
    recurrence; a larger chunk lags the average by one call."""
    cfg = _grid_case(9)
    cfg["model_config"]["load_params"] = {"previous": True}
    cfg["model_config"]["optim_kwargs"] = {"max_iter": 500}                     # run to convergence: optimise_success
    eng = OracleEngine()
    seen = []
    orig = eng.fit_predict_batch

    def spy(**kw):
        seen.append(np.array(kw["theta0"]))
        return orig(**kw)
    eng.fit_predict_batch = spy
    tabs = BatchedLocalExpertOI(engine=eng, **cfg).run(engine_chunk=1)
    assert len(seen) == 9 and all(len(s_) == 1 for s_ in seen)
    ls = tabs["lengthscales"]["lengthscales"].values.reshape(9, 3)
    kv = tabs["kernel_variance"]["kernel_variance"].values
    lv = tabs["likelihood_variance"]["likelihood_variance"].values
    ok = tabs["run_details"]["optimise_success"].values
    prev = np.array([1.0, 1.0, 1.0, 1.0, 1.0])                                  # defaults of the first model
    for t in range(9):
        start = prev.copy()
        start[:3] = np.clip(start[:3], 1e-8 / np.array([0.5, 0.5, 2.0]) + 1e-2, np.array([2.0, 2.0, 4.0]) / np.array([0.5, 0.5, 2.0]) - 1e-2)
        np.testing.assert_allclose(seen[t][0], start, rtol=1e-13)
        if ok[t]:
            prev = 0.95 * prev + 0.05 * np.concatenate([ls[t], [kv[t], lv[t]]])
    assert ok.any()
    # chunk of 4: the tiles of one call share the start of the call
    seen.clear()
    BatchedLocalExpertOI(engine=eng, **cfg).run(engine_chunk=4)
    assert [len(s_) for s_ in seen] == [4, 4, 1]
    assert (seen[0] == seen[0][0]).all() and (seen[1] == seen[1][0]).all() and not np.allclose(seen[0][0], seen[1][0])


def test_explicit_rank_without_group_is_refused_up_front():
    cfg = _grid_case(4)
    eng = OracleEngine()
    with pytest.raises(RuntimeError, match="needs an initialised"):
        BatchedLocalExpertOI(engine=eng, **cfg).run(rank=1, world_size=2)
    assert eng.calls == []                                                      # refused before any work
    part = BatchedLocalExpertOI(engine=eng, **cfg).run(rank=1, world_size=2, gather=False)
    assert 0 < len(part["run_details"]) < 4


def test_prediction_rows_go_out_as_pieces_per_engine_call(tmp_path):
    """A wave's predictions are written as one row piece per engine call while the wave runs (`ResultStore.write_piece`), the
    small tables and the marker at its end; the store reads back exactly what the run returned; pieces of a wave that never
    committed are ignored by readers and removed by the next run; a wave that mixes model profiles is written whole."""
    cfg = _grid_case(9)
    store = str(tmp_path / "s")
    full = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(store_path=store, store_every=8, engine_chunk=3)
    files = sorted(os.listdir(store))
    assert [f for f in files if f.startswith("preds.w000001.")] == [f"preds.w000001.r000.p{j:02d}.parquet" for j in range(3)]
    assert [f for f in files if f.startswith("preds.w000002.")] == ["preds.w000002.r000.p00.parquet"]
    assert "run_details.w000001.r000.parquet" in files and "_wave.w000001.r000.ok" in files
    _assert_same_tables(full, {k: v for k, v in get_results(store).items() if k in full}, ignore=())
    # a fault in the SECOND call of the second wave of a fresh run: wave 1 is committed, wave 2's first piece is on disk, unmarked
    st2 = str(tmp_path / "k")
    with pytest.raises(RuntimeError):
        BatchedLocalExpertOI(engine=DyingEngine(die_at_call=5), **cfg).run(store_path=st2, store_every=6, engine_chunk=2)
    left = sorted(os.listdir(st2))
    assert "_wave.w000001.r000.ok" in left and "_wave.w000002.r000.ok" not in left
    assert any(f.startswith("preds.w000002.r000.p") for f in left)                  # the orphan piece
    assert len(get_results(st2)["run_details"]) == 6                                 # readers see the committed wave only
    n_committed = len(get_results(st2)["preds"])
    assert n_committed == int((full["preds"].index.isin(get_results(st2)["run_details"].index)).sum())
    resumed = BatchedLocalExpertOI(engine=OracleEngine(), **cfg).run(store_path=st2, store_every=6, engine_chunk=2)
    assert len(resumed["run_details"]) == 3                                          # the run continues after the committed wave
    _assert_same_tables(full, {k: v for k, v in get_results(st2, expert_order=True).items() if k in full})
    # two model profiles in one wave (a replacement model below 110 observations): the wave's preds table is written whole
    cfg2 = _grid_case(9)
    cfg2["model_config"]["replacement_threshold"] = 110
    cfg2["model_config"]["replacement_init_params"] = {"kernel": "RBF"}
    st3 = str(tmp_path / "m")
    oi = BatchedLocalExpertOI(engine=OracleEngine(), **cfg2)
    both = oi.run(store_path=st3, store_every=9, engine_chunk=4)
    assert len(set(both["run_details"]["num_obs"] < 110)) == 2                       # both profiles present in the wave
    assert "preds.w000001.r000.parquet" in os.listdir(st3)
    _assert_same_tables(both, {k: v for k, v in get_results(st3).items() if k in both}, ignore=())
