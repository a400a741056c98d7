"""BatchedLocalExpertOI -- the batched counterpart of the per-tile body of the reference's orchestrator.

Reference: ``LocalExpertOI.run`` (GPSat/local_experts.py:761-1279) loops serially over expert locations and, per
tile, selects observations (``DataLoader.local_data_select``, GPSat/dataloader.py:2352-2447), builds the
prediction coordinates (``PredictionLocations``, GPSat/prediction_locations.py:18-43,72-115,208-281), constructs a
model, applies constraints, optimises, evaluates the objective, predicts and appends rows to result tables
(``dict_of_array_to_table``, GPSat/local_experts.py:691-747).

Here the SAME four config dicts are accepted (the in-memory subset listed below), selection is done for ALL
expert locations first (fp64 on the host, the reference's comparison semantics: ``ref + val`` offsets, inclusive
radius for observations, strict ``<`` for prediction locations, source row order preserved), the tiles are packed
into one ragged batch and fitted + predicted by ONE ``gpsat_fit_predict_batch`` call per wave, and the reference's
tables (``run_details``, ``preds``, ``lengthscales``, ``kernel_variance``, ``likelihood_variance``, ``expert_locs``,
``oi_config``; columns ``_dim_0``, ``f*``, ``f*_var``, ``y_var``, ``f_bar``, ``pred_loc_<c>`` ...) are produced with the
expert coordinates as (Multi)Index.  The store is a directory of pandas-pickled tables (pytables/HDF5 is not a dependency
of this backend); re-running skips expert locations already present in ``run_details`` (resume,
local_experts.py:475-497,908-912).

Supported config subset (anything else raises ``NotImplementedError`` -- never a silent fallback):
  expert_loc_config : {"source": DataFrame | csv/parquet path, "sort_by": optional col(s)}
  data_config       : {"data_source": DataFrame | path, "obs_col": str, "coords_col": [..],
                       "local_select": [{"col", "comp", "val"}, ...], "global_select": [static {"col","comp","val"}]}
  pred_loc_config   : {"method": "expert_loc"} | {"method": "from_dataframe", "df": DataFrame, "max_dist": float,
                       "local_select": optional}
  model_config      : {"oi_model": "HipGPRModel" | "GPflowGPRModel" | {"path_to_model", "model_name"},
                       "init_params", "constraints", "optim_kwargs", "pred_kwargs", "params_to_store",
                       "load_params": {"file": store dir | dict of tables, "table_suffix": str}}
``replacement_*`` model settings for tiles below ``replacement_threshold`` observations are honoured (one engine call
per model profile).  ``load_params.previous=True`` (an exponential moving average of earlier tiles' optima, a serial cross-tile
dependency, local_experts.py:1200-1217) is rejected explicitly.
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
from scipy.spatial import cKDTree

from . import _lib as L
from .models import HipGPRModel, LIKELIHOOD_VARIANCE_LOWER_BOUND

_COMPS = {">=": np.greater_equal, ">": np.greater, "==": np.equal, "<": np.less, "<=": np.less_equal}
PARAM_NAMES = ["lengthscales", "kernel_variance", "likelihood_variance"]


# ----------------------------------------------------------------------------------------------------------
# selection (fp64, reference semantics)
# ----------------------------------------------------------------------------------------------------------
def _load_frame(src):
    if isinstance(src, pd.DataFrame):
        return src
    if isinstance(src, str):
        if src.endswith(".parquet"):
            return pd.read_parquet(src)
        if src.endswith(".csv"):
            return pd.read_csv(src)
    raise NotImplementedError(f"data source {type(src)}: this backend takes DataFrames, .csv or .parquet paths")


def data_select(df, selects):
    """Static row selection (GPSat/dataloader.py: data_select with {"col","comp","val"} entries)."""
    keep = np.ones(len(df), dtype=bool)
    for s in selects or []:
        if not {"col", "comp", "val"} <= set(s):
            raise NotImplementedError("only static global_select entries {col, comp, val} are supported")
        assert s["comp"] in _COMPS, f"comp: {s['comp']} is not valid"
        keep &= _COMPS[s["comp"]](df[s["col"]].values, s["val"])
    return df.loc[keep]


class LocalSelector:
    """``DataLoader.local_data_select`` for many reference locations against one frame.

    1-D criteria: ``df[col] <comp> ref[col] + val``; multi-column criteria: Euclidean ball through
    ``KDTree.query_ball_point(x=ref[cols], r=val)`` -- inclusive of points exactly at ``r`` whatever ``comp`` says
    (GPSat/dataloader.py:2413-2444).  The KD-tree is built once (the reference rebuilds it per tile on the same
    frame).  Returns boolean masks, so source row order is kept (dataloader.py:2447)."""

    def __init__(self, df: pd.DataFrame, local_select: List[dict]):
        self.df = df
        self.local_select = local_select or []
        self._trees = {}
        for idx, ls in enumerate(self.local_select):
            col, comp = ls["col"], ls["comp"]
            if isinstance(col, str):
                assert col in df, f"col: {col} is not in data - {df.columns}"
                assert comp in _COMPS, f"comp: {comp} is not valid"
            else:
                assert comp in ["<", "<="], "for multi dimensional values only less than comparison handled"
                for c_ in col:
                    assert c_ in df, f"column: {c_} is not in df.columns: {df.columns}"
                self._trees[idx] = cKDTree(df.loc[:, col].values)

    def mask(self, ref: Dict[str, float]) -> np.ndarray:
        select = np.ones(len(self.df), dtype=bool)
        for idx, ls in enumerate(self.local_select):
            col, comp = ls["col"], ls["comp"]
            if isinstance(col, str):
                assert col in ref, f"col: {col} is not in reference_location - {ref.keys()}"
                select &= _COMPS[comp](self.df[col].values, ref[col] + ls["val"])
            else:
                for c_ in col:
                    assert c_ in ref, f"col: {col} is not in reference_location - {ref.keys()}"
                ids = self._trees[idx].query_ball_point(x=[ref[c_] for c_ in col], r=ls["val"])
                m = np.zeros(len(self.df), dtype=bool)
                m[ids] = True
                select &= m
        return select


class DeviceSelector:
    """The same membership as ``LocalSelector`` / ``max_dist_bool`` for ALL reference locations in one GPU call
    (``gpsat_select_batch``: fp64 predicates with the reference's arithmetic, bit-exact, source row order).

    ``local_select`` entries as in the reference; ``strict_ball=True`` gives the prediction-location semantics
    (strict ``<`` on the squared distance, GPSat/prediction_locations.py:37,43)."""

    def __init__(self, df: pd.DataFrame, local_select: List[dict], engine, strict_ball: bool = False):
        self.engine = engine
        cols = []
        for ls in local_select:
            for c_ in ([ls["col"]] if isinstance(ls["col"], str) else list(ls["col"])):
                assert c_ in df, f"column: {c_} is not in df.columns: {df.columns}"
                if c_ not in cols:
                    cols.append(c_)
        self.cols = cols
        self.points = df.loc[:, cols].values.astype(np.float64)
        self.criteria = []
        for ls in local_select:
            if isinstance(ls["col"], str):
                assert ls["comp"] in _COMPS, f"comp: {ls['comp']} is not valid"
                self.criteria.append(("cmp", cols.index(ls["col"]), ls["comp"], ls["val"]))
            else:
                assert ls["comp"] in ["<", "<="], "for multi dimensional values only less than comparison handled"
                if len(ls["col"]) > 3:
                    raise NotImplementedError("device ball selection takes 1..3 columns")
                self.criteria.append(("ball", [cols.index(c_) for c_ in ls["col"]], "<" if strict_ball else "<=", ls["val"]))

    def select(self, refs: pd.DataFrame):
        """refs: one row per expert with (at least) the columns used by the criteria.
        Returns CSR (off [T+1], idx [off[-1]]) of selected row POSITIONS of the frame, ascending per expert."""
        for c_ in self.cols:
            assert c_ in refs, f"col: {c_} is not in reference_location - {list(refs.columns)}"
        return self.engine.select_batch(self.points, refs.loc[:, self.cols].values.astype(np.float64), self.criteria)


def max_dist_bool(loc: np.ndarray, ref_loc: np.ndarray, max_dist: float) -> np.ndarray:
    """``_max_dist_bool`` (GPSat/prediction_locations.py:18-43): per-dimension pre-filter, then the STRICT test
    sum((loc - ref)^2) < max_dist^2, in fp64."""
    d = loc - ref_loc[None, :]
    m2 = max_dist * max_dist
    out = np.all(d * d < m2, axis=1)
    d2 = np.sum(d ** 2, axis=1)
    return out & (d2 < m2)


class PredictionLocations:
    """Prediction coordinates per expert (GPSat/prediction_locations.py:72-115,208-281): ``expert_loc`` or
    ``from_dataframe`` with ``max_dist`` over the columns present in the frame; dimensions missing from the frame
    are filled from the expert location (:262-271); optional ``local_select`` afterwards (:106-111)."""

    def __init__(self, method="expert_loc", coords_col=None, df=None, max_dist=None, local_select=None, **kw):
        if method not in ("expert_loc", "from_dataframe"):
            raise NotImplementedError(f"prediction location method '{method}' is not built (expert_loc, from_dataframe)")
        if kw:
            raise NotImplementedError(f"unsupported pred_loc_config keys: {sorted(kw)}")
        self.method, self.coords_col, self.max_dist, self.local_select = method, list(coords_col), max_dist, local_select
        if method == "from_dataframe":
            df = _load_frame(df)
            self.found = [c for c in self.coords_col if c in df.columns]
            self.fc_loc = [self.coords_col.index(c) for c in self.found]
            self.vals = df.loc[:, self.found].values.astype(np.float64)
            if local_select:
                self._frame = pd.DataFrame(self.vals, columns=self.found)

    def __call__(self, expert_loc: np.ndarray) -> np.ndarray:
        """expert_loc: (D,) fp64 in the order of coords_col.  Returns (P, D) fp64."""
        if self.method == "expert_loc":
            return expert_loc[None, :].copy()
        b = max_dist_bool(self.vals, expert_loc[self.fc_loc], self.max_dist) if self.max_dist is not None \
            else np.ones(len(self.vals), dtype=bool)
        if self.local_select:
            ref = {c: expert_loc[i] for i, c in enumerate(self.coords_col)}
            b = b & LocalSelector(self._frame, self.local_select).mask(ref)
        out = np.full((int(b.sum()), len(self.coords_col)), np.nan)
        out[:, self.fc_loc] = self.vals[b]
        missing = [i for i, c in enumerate(self.coords_col) if c not in self.found]
        out[:, missing] = expert_loc[missing]
        return out


# ----------------------------------------------------------------------------------------------------------
# result tables (GPSat/local_experts.py:691-747, GPSat/utils.py:1437-1495)
# ----------------------------------------------------------------------------------------------------------
def _index_for(coords_col, loc_rows: np.ndarray):
    if len(coords_col) == 1:
        return pd.Index(loc_rows[:, 0], name=coords_col[0])
    return pd.MultiIndex.from_arrays([loc_rows[:, i] for i in range(loc_rows.shape[1])], names=coords_col)


class ResultStore:
    """Directory of pandas-pickled tables named ``<table><suffix>.pkl`` (+ ``oi_config<suffix>.json``): no
    dependency beyond pandas itself (pytables / pyarrow are not guaranteed on the GPU hosts)."""

    def __init__(self, path: Optional[str]):
        self.path = path
        if path:
            os.makedirs(path, exist_ok=True)

    def _file(self, table):
        return os.path.join(self.path, f"{table}.pkl")

    def read(self, table) -> Optional[pd.DataFrame]:
        if not self.path or not os.path.exists(self._file(table)):
            return None
        return pd.read_pickle(self._file(table))

    def append(self, table, df: pd.DataFrame):
        if not self.path or df is None or len(df) == 0:
            return
        old = self.read(table)
        if old is not None:
            df = pd.concat([old, df])
        df.to_pickle(self._file(table))

    def tables(self) -> Dict[str, pd.DataFrame]:
        out = {}
        if self.path:
            for f in sorted(os.listdir(self.path)):
                if f.endswith(".pkl"):
                    out[f[:-4]] = pd.read_pickle(os.path.join(self.path, f))
        return out


def get_results(store_path: str) -> Dict[str, pd.DataFrame]:
    """Counterpart of ``get_results_from_h5file`` (GPSat/local_experts.py:1467): all tables of a store."""
    return ResultStore(store_path).tables()


# ----------------------------------------------------------------------------------------------------------
# the batched orchestrator
# ----------------------------------------------------------------------------------------------------------
class BatchedLocalExpertOI:
    def __init__(self, expert_loc_config: dict, data_config: dict, model_config: dict, pred_loc_config: dict,
                 engine=None, device_select: bool = False):
        self.config = {"locations": _jsonable(expert_loc_config), "data": _jsonable(data_config),
                       "model": _jsonable(model_config), "pred_loc": _jsonable(pred_loc_config)}
        # ---- data (local_experts.py:266-290)
        self.obs_col = data_config["obs_col"]
        self.coords_col = list(data_config["coords_col"])
        if isinstance(self.obs_col, (list, tuple)):
            assert len(self.obs_col) == 1
            self.obs_col = self.obs_col[0]
        self.local_select = data_config.get("local_select", [])
        df = data_select(_load_frame(data_config["data_source"]), data_config.get("global_select"))
        self.df = df
        # ---- expert locations (local_experts.py:349-422)
        xl = _load_frame(expert_loc_config["source"])
        if expert_loc_config.get("sort_by") is not None:
            xl = xl.sort_values(expert_loc_config["sort_by"])
        for k in expert_loc_config:
            if k not in ("source", "sort_by"):
                raise NotImplementedError(f"expert_loc_config key '{k}' is not supported by the batched backend")
        self.expert_locs = xl.reset_index(drop=True)
        # ---- model (local_experts.py:292-346)
        om = model_config.get("oi_model", "HipGPRModel")
        name = om["model_name"] if isinstance(om, dict) else om
        if name not in ("HipGPRModel", "GPflowGPRModel"):
            raise NotImplementedError(f"oi_model '{name}': the batched backend builds the exact-GP expert only")
        self.init_params = dict(model_config.get("init_params") or {})
        self.constraints = model_config.get("constraints")
        self.optim_kwargs = dict(model_config.get("optim_kwargs") or {})
        self.pred_kwargs = dict(model_config.get("pred_kwargs") or {})
        # replacement model for tiles with fewer than `replacement_threshold` observations (local_experts.py:339-346,
        # 1021-1041): same defaults as the reference -- init_params / constraints fall back to the main ones,
        # optim_kwargs / pred_kwargs to {}
        self.replacement_threshold = model_config.get("replacement_threshold")
        self.profiles = {"main": dict(init_params=self.init_params, constraints=self.constraints,
                                      optim_kwargs=self.optim_kwargs, pred_kwargs=self.pred_kwargs)}
        if self.replacement_threshold is not None:
            rm = model_config.get("replacement_model")
            rname = rm["model_name"] if isinstance(rm, dict) else rm
            if rname not in (None, "HipGPRModel", "GPflowGPRModel"):
                raise NotImplementedError(f"replacement_model '{rname}': the batched backend builds the exact-GP expert only")
            rip = model_config.get("replacement_init_params")
            rco = model_config.get("replacement_constraints")
            self.profiles["replacement"] = dict(
                init_params=self.init_params if rip is None else dict(rip),
                constraints=self.constraints if rco is None else rco,
                optim_kwargs=dict(model_config.get("replacement_optim_kwargs") or {}),
                pred_kwargs=dict(model_config.get("replacement_pred_kwargs") or {}))
        for pf in self.profiles.values():
            if pf["pred_kwargs"].get("full_cov"):
                raise NotImplementedError("full_cov=True tables are not written by the batched orchestrator "
                                          "(use HipGPRModel.predict(full_cov=True) / Engine.fit_predict_batch(full_cov=True))")
        self.params_to_store = model_config.get("params_to_store") or PARAM_NAMES
        self.load_params = model_config.get("load_params")
        if self.load_params is not None and self.load_params.get("previous", False):
            raise NotImplementedError("load_params.previous=True is a serial cross-tile dependency "
                                      "(local_experts.py:1200-1217) and is not defined for batched execution")
        # ---- prediction locations (local_experts.py:254-264)
        plc = dict(pred_loc_config or {"method": "expert_loc"})
        self.pred_loc = PredictionLocations(coords_col=self.coords_col, **plc)
        from .engine import default_engine
        self.engine = engine if engine is not None else default_engine()
        # tile membership for all experts in one GPU call (bit-identical to the host selector)
        self.device_select = device_select

    # -- per-tile host-side model logic reuses the drop-in class (intake, scaling, defaults, constraints)
    def _host_model(self, df_local, init_params=None):
        return HipGPRModel(data=df_local, obs_col=self.obs_col, coords_col=self.coords_col, engine=self.engine,
                           verbose=False, **(self.init_params if init_params is None else init_params))

    def _loaded_params(self, store: ResultStore, suffix, ref_row, model):
        """load_params from tables ``<param><suffix>`` where the index equals the expert coordinates
        (local_experts.py:553-689).  Returns False when nothing usable was found (tile is skipped)."""
        lp = self.load_params
        src = lp.get("file")
        tsuf = lp.get("table_suffix", suffix)
        found = False
        for pn in lp.get("param_names", PARAM_NAMES):
            tab = src.get(f"{pn}{tsuf}") if isinstance(src, dict) else ResultStore(src).read(f"{pn}{tsuf}")
            if tab is None:
                continue
            key = tuple(ref_row[c] for c in self.coords_col)
            try:
                rows = tab.loc[[key if len(key) > 1 else key[0]]]
            except KeyError:
                continue
            vals = rows.sort_values("_dim_0")[pn].values.astype(np.float64)
            if len(vals) == 0 or np.isnan(vals).any():
                continue                                   # NaN -> parameter dropped (local_experts.py:670-679)
            model.set_parameters(**{pn: vals if pn == "lengthscales" else float(vals[0])})
            found = True
        return found

    def run(self, store_path: Optional[str] = None, optimise: bool = True, predict: bool = True, min_obs: int = 3,
            table_suffix: str = "", max_tiles_per_call: Optional[int] = None, check_config_compatible: bool = True):
        t_start = time.perf_counter()
        store = ResultStore(store_path)
        cc = self.coords_col
        xl = self.expert_locs
        # expert_locs table + config bookkeeping (local_experts.py:873-903)
        if store_path:
            cfg_file = os.path.join(store_path, f"oi_config{table_suffix}.json")
            prev = json.load(open(cfg_file)) if os.path.exists(cfg_file) else []
            prev.append({"idx": len(prev) + 1, "datetime": time.strftime("%Y-%m-%d %H:%M:%S"), "config": self.config})
            json.dump(prev, open(cfg_file, "w"))
            config_id = len(prev)
            if store.read(f"expert_locs{table_suffix}") is None:
                store.append(f"expert_locs{table_suffix}", xl.set_index(cc))
        else:
            config_id = 1
        # resume: drop expert locations already in run_details (local_experts.py:475-497,908-912)
        todo = np.ones(len(xl), dtype=bool)
        done = store.read(f"run_details{table_suffix}")
        if done is not None and len(done):
            have = set(done.index.tolist())
            keys = [tuple(r) if len(cc) > 1 else r[0] for r in xl[cc].values.tolist()]
            todo = np.array([k not in have for k in keys])
        selector = LocalSelector(self.df, self.local_select)
        dev_off = dev_idx = None
        if self.device_select and len(self.local_select):
            dev_off, dev_idx = DeviceSelector(self.df, self.local_select, self.engine).select(xl)
        D = len(cc)
        # per profile (main / replacement): what one engine call needs to be uniform in
        prof = {}
        for pname, pf in self.profiles.items():
            kernel = pf["init_params"].get("kernel", "Matern32")
            if kernel not in L.KERNEL_IDS:
                raise NotImplementedError(f"kernel {kernel!r}")
            ok = pf["optim_kwargs"]
            prof[pname] = dict(kernel=kernel, fixed=list(ok.get("fixed_params") or []), max_iter=int(ok.get("max_iter", 10_000)),
                               eng_kw={k: ok[k] for k in ("max_ls", "ftol", "gtol", "adam_lr") if k in ok},
                               optimiser=ok.get("optimiser", "lbfgs") if optimise else "none", **pf)

        # ---------------- pass 1: selection + host-side model logic for every expert (fp64) ----------------
        tiles, stubs = [], []
        for i in np.nonzero(todo)[0]:
            rl = xl.iloc[i]
            ref = {c: rl[c] for c in xl.columns}
            loc = rl[cc].values.astype(np.float64)
            pc = self.pred_loc(loc)
            if len(pc) == 0:                                   # local_experts.py:962-965: skipped, nothing stored
                continue
            df_local = self.df.iloc[dev_idx[dev_off[i]:dev_off[i + 1]]] if dev_idx is not None \
                else self.df.loc[selector.mask(ref)]
            if len(df_local) < min_obs:                        # local_experts.py:988-1012: stub run_details row
                stubs.append((loc, len(df_local)))
                continue
            pname = "replacement" if (self.replacement_threshold is not None and
                                      len(df_local) < self.replacement_threshold) else "main"   # local_experts.py:1021-1041
            pf = prof[pname]
            m = self._host_model(df_local, pf["init_params"])
            save_params = True
            if self.load_params is not None:
                if not self._loaded_params(store, table_suffix, ref, m):
                    continue                                   # local_experts.py:1099-1101
                same = (self.load_params.get("file") == store_path and
                        self.load_params.get("table_suffix", table_suffix) == table_suffix and
                        set(self.load_params) <= {"file", "table_suffix"})
                save_params = not (same and not optimise)      # local_experts.py:1090-1097
            if pf["constraints"] is not None:
                cons = {k: dict(v) for k, v in pf["constraints"].items()}
                if pf["init_params"].get("coords_scale", None) is not None and "lengthscales" in cons:
                    cons["lengthscales"]["scale"] = True       # local_experts.py:1113-1114
                m.set_parameter_constraints(cons, move_within_tol=True, tol=1e-2)
            m._fix_hyperparameters(pf["fixed"])
            pcs = pc / m.coords_scale if pf["pred_kwargs"].get("apply_scale", True) else pc
            tiles.append(dict(loc=loc, model=m, pred_raw=pc, pred_scaled=pcs, save_params=save_params, profile=pname))

        # ---------------- pass 2: one packed batch per wave (and per model profile) through the C ABI ----------------
        out = {k: [] for k in ("run_details", "preds", *self.params_to_store)}
        res = [None] * len(tiles)                 # per tile: (theta, nll, status, f_mean, f_var, y_var, seconds)
        for pname, pf in prof.items():
            members = [i for i, t in enumerate(tiles) if t["profile"] == pname]
            wave = max_tiles_per_call or max(len(members), 1)
            for w0 in range(0, len(members), wave):
                ids = members[w0:w0 + wave]
                tw = [tiles[i] for i in ids]
                t0 = time.perf_counter()
                Ns = np.array([len(t["model"].coords) for t in tw])
                Ps = np.array([len(t["pred_scaled"]) if predict else 0 for t in tw])
                obs_off = np.concatenate([[0], np.cumsum(Ns)])
                pred_off = np.concatenate([[0], np.cumsum(Ps)])
                X = np.concatenate([t["model"].coords for t in tw]).astype(np.float32)
                y = np.concatenate([t["model"].obs[:, 0] for t in tw]).astype(np.float32)
                Xs = np.concatenate([t["pred_scaled"] if predict else np.zeros((0, D)) for t in tw]).astype(np.float32)
                theta0 = np.stack([t["model"]._theta for t in tw])
                lo = np.stack([t["model"]._lo for t in tw])
                hi = np.stack([t["model"]._hi for t in tw])
                trainable = tw[0]["model"]._trainable
                r = self.engine.fit_predict_batch(D=D, obs_off=obs_off, X=X, y=y, pred_off=pred_off, Xs=Xs, theta0=theta0,
                                                  lo=lo, hi=hi, trainable=trainable, kernel=pf["kernel"],
                                                  optimiser=pf["optimiser"], max_iter=pf["max_iter"], **pf["eng_kw"])
                dt = (time.perf_counter() - t0) / max(len(tw), 1)
                for k, i in enumerate(ids):
                    a_, b_ = pred_off[k], pred_off[k + 1]
                    res[i] = (r.theta[k], float(r.nll[k]), int(r.status[k]), r.f_mean[a_:b_], r.f_var[a_:b_], r.y_var[a_:b_], dt)
        for t, rr in zip(tiles, res):              # tables in expert order, whatever the grouping above
            th, nll_k, st_k, fm, fv, yv, dt = rr
            m = t["model"]
            loc = t["loc"]
            idx1 = _index_for(cc, loc[None, :])
            out["run_details"].append(pd.DataFrame({
                "_dim_0": [0], "num_obs": [len(m.coords)], "run_time": [dt], "objective_value": [nll_k],
                "parameters_optimised": [bool(optimise)], "optimise_success": [bool(optimise and st_k == 0)],
                "model": [f"{HipGPRModel.__module__}.{HipGPRModel.__name__}"[:64]],
                "device": [str(m.gpu_name)[:64]], "config_id": [config_id]}, index=idx1))
            if t["save_params"]:
                vals = {"lengthscales": th[:D], "kernel_variance": th[D:D + 1], "likelihood_variance": th[D + 1:D + 2]}
                for pn in self.params_to_store:
                    v = np.asarray(vals[pn], dtype=np.float64)
                    out[pn].append(pd.DataFrame({"_dim_0": np.arange(len(v)), pn: v},
                                                index=_index_for(cc, np.repeat(loc[None, :], len(v), 0))))
            P = len(fm) if predict else 0
            if P > 0:
                pr = {"_dim_0": np.arange(P), "f*": np.asarray(fm, dtype=np.float64),
                      "f*_var": np.asarray(fv, dtype=np.float64), "y_var": np.asarray(yv, dtype=np.float64),
                      "f_bar": np.repeat(m.obs_mean[:, 0], P)}
                for ci, c_ in enumerate(cc):
                    pr[f"pred_loc_{c_}"] = t["pred_raw"][:, ci]
                out["preds"].append(pd.DataFrame(pr, index=_index_for(cc, np.repeat(loc[None, :], P, 0))))
        for loc, n in stubs:
            out["run_details"].append(pd.DataFrame({
                "_dim_0": [0], "num_obs": [int(n)], "run_time": [np.nan], "objective_value": [np.nan],
                "parameters_optimised": [bool(optimise)], "optimise_success": [False],
                "model": [f"{HipGPRModel.__module__}.{HipGPRModel.__name__}"[:64]], "device": [""],
                "config_id": [config_id]}, index=_index_for(cc, loc[None, :])))
        tables = {f"{k}{table_suffix}": (pd.concat(v) if len(v) else pd.DataFrame()) for k, v in out.items()}
        for k, v in tables.items():
            store.append(k, v)
        self.run_seconds = time.perf_counter() - t_start
        return tables


def _jsonable(cfg):
    def conv(v):
        if isinstance(v, pd.DataFrame):
            return f"<DataFrame {v.shape[0]}x{v.shape[1]}>"
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [conv(x) for x in v]
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, (np.integer, np.floating)):
            return v.item()
        return v
    return conv(cfg or {})
