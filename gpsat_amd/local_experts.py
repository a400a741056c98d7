"""BatchedLocalExpertOI -- the batched counterpart of the per-tile body of the reference's orchestrator.

Reference: ``LocalExpertOI.run`` (GPSat/local_experts.py:761-1279) loops serially over expert locations and, per
tile, selects observations (``DataLoader.local_data_select``, GPSat/dataloader.py:2352-2447), builds the
prediction coordinates (``PredictionLocations``, GPSat/prediction_locations.py:18-43,72-115,208-281), constructs a
model, applies constraints, optimises, evaluates the objective, predicts and appends rows to result tables
(``dict_of_array_to_table``, GPSat/local_experts.py:691-747) every ``store_every`` tiles (:500-548).

Here the SAME four config dicts are accepted (the in-memory subset listed below).  The work is organised as

  pass 1  tile membership for ALL expert locations (fp64, the reference's comparison semantics: ``ref + val``
          offsets, inclusive radius for observations, strict ``<`` for prediction locations, source row order;
          one GPU call with ``device_select=True``), scaling / de-meaning and the per-tile parameter vectors
          (defaults, loaded parameters, box constraints with move-within-tol) as whole-array operations -- no per-tile
          model object, no per-tile DataFrame;
  shard   with ``world_size > 1`` the global tile list is split by ``sharding.partition_tiles`` (LPT on
          E*N^3 + N^2*P); every rank packs and runs only its own tiles; no data-path collective;
  pass 2  waves of ``store_every`` expert locations: one ``gpsat_fit_predict_batch`` call per wave (and per model
          profile), tables assembled with array operations and flushed to the store as one append-only, atomically
          committed part per wave -- a fault at tile 99 999 loses at most the running wave, and a re-run resumes
          after the last committed wave (the reference's resume contract, local_experts.py:475-497,908-912);
  gather  with ``world_size > 1`` ONE gather (``sharding.gather_results``: RCCL over xGMI on the GPU node) returns
          per-tile hyper-parameters + predictions to rank 0, which assembles the tables in the reference's expert order.

Tables (``run_details``, ``preds``, ``lengthscales``, ``kernel_variance``, ``likelihood_variance``, ``expert_locs``,
``oi_config``; columns ``_dim_0``, ``f*``, ``f*_var``, ``y_var``, ``f_bar``, ``pred_loc_<c>`` ...) carry the expert
coordinates as (Multi)Index like the reference's.  The store is a directory of Apache Parquet parts (pandas pickles
when pyarrow is absent; pytables / HDF5 is not a dependency of this backend -- ``export_parquet`` / ``export_hdf5``).

Supported config subset (anything else raises ``NotImplementedError`` -- never a silent fallback):
  expert_loc_config : {"source": DataFrame | csv/parquet path, "sort_by": optional col(s)}
  data_config       : {"data_source": DataFrame | path, "obs_col": str, "coords_col": [..],
                       "local_select": [{"col", "comp", "val"}, ...], "global_select": [static {"col","comp","val"}]}
  pred_loc_config   : {"method": "expert_loc"} | {"method": "from_dataframe", "df": DataFrame, "max_dist": float,
                       "local_select": optional} | {"method": "from_source", "load_kwargs": {"source": DataFrame | path},
                       "max_dist": ..} | {"method": "shift_arrays", "<coord>": array, ...}
  model_config      : {"oi_model": "HipGPRModel" | "GPflowGPRModel" | {"path_to_model", "model_name"},
                       "init_params", "constraints", "optim_kwargs", "pred_kwargs", "params_to_store",
                       "load_params": {"file": store dir | dict of tables, "table_suffix": str, "param_names": [..],
                                       "index_adjust": {col: {"func": callable | "lambda ..."}}} |
                                      {<param>: value, ...}  (set directly on every tile)}
``replacement_*`` model settings for tiles below ``replacement_threshold`` observations are honoured (one engine call
per model profile and wave).  ``pred_kwargs.full_cov=True`` adds the table ``preds_2`` (``_dim_0``, ``_dim_1``, ``f*_cov``,
``y_cov``: what ``dict_of_array_to_table(concat=True, table="preds")`` makes of the 2-D arrays of the prediction dict,
local_experts.py:691-747, gpflow_models.py:245-263).
``load_params.previous=True`` (local_experts.py:1059-1064,1200-1217: every tile starts from an exponential moving average,
rho = 0.95, of the optima of the successfully optimised tiles before it) is a serial cross-tile dependency; its batched
definition here is CALL-LAGGED: all tiles of one engine call start from the average as it stood when the call was
issued, and the average is then advanced over that call's tiles in expert order.  ``engine_chunk=1`` reproduces the
reference's serial recurrence exactly; larger chunks trade its freshness for batching (sharded runs keep one average
per rank).
"""
from __future__ import annotations

import json
import os
import re
import time
import warnings
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
from scipy.spatial import cKDTree

from . import _lib as L
from . import sharding
from .models import HipGPRModel, LIKELIHOOD_VARIANCE_LOWER_BOUND, clamp_within

_COMPS = {">=": np.greater_equal, ">": np.greater, "==": np.equal, "<": np.less, "<=": np.less_equal}
PARAM_NAMES = ["lengthscales", "kernel_variance", "likelihood_variance"]
MODEL_NAME = f"{HipGPRModel.__module__}.{HipGPRModel.__name__}"[:64]
DTYPES = ("f32", "f64")


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
        self._cols = {}
        for idx, ls in enumerate(self.local_select):
            col, comp = ls["col"], ls["comp"]
            if isinstance(col, str):
                assert col in df, f"col: {col} is not in data - {df.columns}"
                assert comp in _COMPS, f"comp: {comp} is not valid"
                self._cols[col] = df[col].values
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
                select &= _COMPS[comp](self._cols[col], ref[col] + ls["val"])
            else:
                for c_ in col:
                    assert c_ in ref, f"col: {col} is not in reference_location - {ref.keys()}"
                ids = self._trees[idx].query_ball_point(x=[ref[c_] for c_ in col], r=ls["val"])
                m = np.zeros(len(self.df), dtype=bool)
                m[ids] = True
                select &= m
        return select

    def select(self, refs: pd.DataFrame):
        """CSR (off [T+1], idx) of selected row POSITIONS for every row of ``refs`` (same layout as DeviceSelector)."""
        cols = list(refs.columns)
        vals = refs.values
        chunks, off = [], np.zeros(len(refs) + 1, dtype=np.int64)
        for i in range(len(refs)):
            ids = np.nonzero(self.mask(dict(zip(cols, vals[i]))))[0]
            chunks.append(ids)
            off[i + 1] = off[i] + len(ids)
        return off, (np.concatenate(chunks) if chunks else np.zeros(0, np.int64)).astype(np.int64)


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
        self.points_cm = np.ascontiguousarray(self.points.T)          # what the C ABI takes: column-major, made once
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
        return self.engine.select_batch(None, refs.loc[:, self.cols].values.astype(np.float64), self.criteria,
                                        points_cm=self.points_cm)


def max_dist_bool(loc: np.ndarray, ref_loc: np.ndarray, max_dist: float) -> np.ndarray:
    """``_max_dist_bool`` (GPSat/prediction_locations.py:18-43): per-dimension pre-filter, then the STRICT test
    sum((loc - ref)^2) < max_dist^2, in fp64."""
    d = loc - ref_loc[None, :]
    m2 = max_dist * max_dist
    out = np.all(d * d < m2, axis=1)
    d2 = np.sum(d ** 2, axis=1)
    return out & (d2 < m2)


class PredictionLocations:
    """Prediction coordinates per expert (GPSat/prediction_locations.py:72-115,182-281):

    ``expert_loc``      the expert location itself;
    ``shift_arrays``    the mesh of per-coordinate shift arrays (``<coord>=array`` keywords, missing coordinates shift
                        by 0; built once, :182-206) added to the expert location;
    ``from_dataframe``  rows of a frame within ``max_dist`` (strict) over the columns present in the frame; dimensions
                        missing from the frame are filled from the expert location (:262-271); optional
                        ``local_select`` afterwards (:106-111);
    ``from_source``     ``load_kwargs={"source": DataFrame | csv | parquet}`` loaded once, duplicates dropped, then as
                        ``from_dataframe`` (:83-101)."""

    def __init__(self, method="expert_loc", coords_col=None, df=None, max_dist=None, local_select=None,
                 load_kwargs=None, Xout=None, **kw):
        self.coords_col = list(coords_col)
        if method == "from_source":
            assert load_kwargs is not None, \
                "calling PredictionLocations object with method='from_source', however 'load_kwargs' is missing"
            extra = set(load_kwargs) - {"source"}
            if extra:
                raise NotImplementedError(f"from_source load_kwargs {sorted(extra)}: only 'source' is supported")
            df = _load_frame(load_kwargs["source"]).drop_duplicates()
            method = "from_dataframe"
        if method not in ("expert_loc", "from_dataframe", "shift_arrays"):
            raise ValueError(f"prediction location method '{method}' is not implemented")
        self.method, self.max_dist, self.local_select = method, max_dist, local_select
        if method == "shift_arrays":
            unknown = set(kw) - set(self.coords_col)
            if unknown:
                raise NotImplementedError(f"shift_arrays: {sorted(unknown)} are not coordinate columns")
            if Xout is None:
                axes = [np.atleast_1d(np.asarray(kw.get(c, np.zeros(1)), dtype=np.float64)) for c in self.coords_col]
                for a in axes:
                    assert a.ndim == 1
                mesh = np.meshgrid(*axes, indexing="ij")
                Xout = np.stack([m.reshape(-1) for m in mesh], axis=1)
            self.shifts = np.asarray(Xout, dtype=np.float64)
            assert self.shifts.ndim == 2 and self.shifts.shape[1] == len(self.coords_col)
        elif kw:
            raise NotImplementedError(f"unsupported pred_loc_config keys: {sorted(kw)}")
        if method == "from_dataframe":
            df = _load_frame(df)
            self.found = [c for c in self.coords_col if c in df.columns]
            self.fc_loc = [self.coords_col.index(c) for c in self.found]
            self.vals = df.loc[:, self.found].values.astype(np.float64)
            self.missing = [i for i, c in enumerate(self.coords_col) if c not in self.found]
            self._sel = LocalSelector(pd.DataFrame(self.vals, columns=self.found), local_select) if local_select else None

    def __call__(self, expert_loc: np.ndarray) -> np.ndarray:
        """expert_loc: (D,) fp64 in the order of coords_col.  Returns (P, D) fp64."""
        if self.method == "expert_loc":
            return expert_loc[None, :].copy()
        if self.method == "shift_arrays":
            return self.shifts + expert_loc[None, :]
        b = max_dist_bool(self.vals, expert_loc[self.fc_loc], self.max_dist) if self.max_dist is not None \
            else np.ones(len(self.vals), dtype=bool)
        if self._sel is not None:
            b = b & self._sel.mask({c: expert_loc[i] for i, c in enumerate(self.coords_col)})
        return self._rows(np.nonzero(b)[0], expert_loc)

    def _rows(self, ids, expert_loc):
        out = np.empty((len(ids), len(self.coords_col)))
        out[:, self.fc_loc] = self.vals[ids]
        out[:, self.missing] = expert_loc[self.missing]
        return out

    def batch(self, locs: np.ndarray, engine=None):
        """All experts at once: a ``RaggedRows`` (``[i]`` is expert i's (P_i, D) array).  With an engine the ``max_dist``
        filter of ``from_dataframe`` runs as ONE ``gpsat_select_batch`` call (strict ball, bit-identical membership) and the
        coordinates are gathered in one piece (a Python-level gather per expert cost 4 us x 16 384 experts = 0.07 s)."""
        D = len(self.coords_col)
        if self.method == "from_dataframe" and engine is not None and self.max_dist is not None and self._sel is None \
                and 1 <= len(self.found) <= 3:
            frame = pd.DataFrame(self.vals, columns=self.found)
            ds = DeviceSelector(frame, [{"col": list(self.found), "comp": "<", "val": self.max_dist}], engine,
                                strict_ball=True)
            off, idx = ds.select(pd.DataFrame(locs[:, self.fc_loc], columns=self.found))
            cat = np.empty((len(idx), D))
            cat[:, self.fc_loc] = self.vals[idx]
            if self.missing:
                cat[:, self.missing] = np.repeat(locs[:, self.missing], np.diff(off), axis=0)
            return RaggedRows(cat, off)
        rows = [self(locs[i]) for i in range(len(locs))]
        off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
        return RaggedRows(np.concatenate(rows) if len(rows) else np.zeros((0, D)), off)


class RaggedRows:
    """Row blocks of unequal length kept back to back: ``cat`` (sum P_i, D) and ``off`` (T + 1).  Reads like a list of
    arrays; ``take`` gathers the blocks of many items without a Python loop."""

    def __init__(self, cat: np.ndarray, off: np.ndarray):
        self.cat, self.off = cat, np.asarray(off, dtype=np.int64)
        self.counts = np.diff(self.off)

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        return self.cat[self.off[i]:self.off[i + 1]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def take(self, items) -> np.ndarray:
        """The blocks of ``items`` (positions, any order) back to back."""
        items = np.asarray(items, dtype=np.int64)
        if len(items) == 0:
            return self.cat[:0]
        if bool(np.all(np.diff(items) == 1)):
            return self.cat[self.off[items[0]]:self.off[items[-1] + 1]]
        cnt = self.counts[items]
        tot = int(cnt.sum())
        start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        rows = np.arange(tot) - np.repeat(start, cnt) + np.repeat(self.off[items], cnt)
        return self.cat[rows]


# ----------------------------------------------------------------------------------------------------------
# result tables (GPSat/local_experts.py:691-747, GPSat/utils.py:1437-1495) and their store
# ----------------------------------------------------------------------------------------------------------
def _index_for(coords_col, loc_rows: np.ndarray):
    if len(coords_col) == 1:
        return pd.Index(loc_rows[:, 0], name=coords_col[0])
    return pd.MultiIndex.from_arrays([loc_rows[:, i] for i in range(loc_rows.shape[1])], names=coords_col)


def _index_for_repeated(coords_col, loc_rows: np.ndarray, counts):
    """``_index_for(coords_col, np.repeat(loc_rows, counts, axis=0))`` without factorising the repeated rows: the levels come
    from the (few) distinct locations, the codes are repeated (a tenth of the time for the 210 k rows of a preds table)."""
    counts = np.asarray(counts)
    if len(coords_col) == 1 or len(loc_rows) == 0:
        return _index_for(coords_col, np.repeat(loc_rows, counts, axis=0))
    levels, codes = [], []
    for i in range(loc_rows.shape[1]):
        c_, l_ = pd.factorize(loc_rows[:, i], sort=True)
        levels.append(l_)
        codes.append(np.repeat(c_, counts))
    return pd.MultiIndex(levels=levels, codes=codes, names=coords_col, verify_integrity=False)


_PART_RE = re.compile(r"^(?P<table>[^.].*)\.w(?P<k>\d{6})\.r(?P<r>\d{3})(\.p(?P<piece>\d{2}))?\.(?P<ext>parquet|pkl)$")
_PIECE_ROWS = 65536            # a table of a wave longer than this is written as up to _PIECE_MAX row pieces, in parallel
_PIECE_MAX = 4
_writers = []


def _writer_pool():
    if not _writers:
        from concurrent.futures import ThreadPoolExecutor
        _writers.append(ThreadPoolExecutor(max_workers=_PIECE_MAX))
    return _writers[0]
_MARK_RE = re.compile(r"^_wave\.w(?P<k>\d{6})\.r(?P<r>\d{3})\.ok$")
_WHOLE_RE = re.compile(r"^(?P<table>[^.].*)\.(?P<ext>parquet|pkl)$")


def _have_pyarrow():
    try:
        import pyarrow  # noqa: F401
        return True
    except Exception:
        return False


def _read_part(path):
    if path.endswith(".parquet"):
        df = pd.read_parquet(path)
        # pandas keeps the (Multi)Index in the parquet metadata; a frame written without rows comes back the same way
        return df
    return pd.read_pickle(path)


class ResultStore:
    """Directory store of table parts: Apache Parquet (pyarrow) by default -- readable by ``pandas.read_parquet`` /
    pyarrow / any parquet tool, independent of the pandas version, safe to load -- or pandas pickles (``fmt="pickle"``,
    and the fallback when pyarrow is absent; stores written by earlier versions stay readable).  HDF5 / pytables, the
    reference's container, is not a dependency of this backend (``export_hdf5`` converts when pytables is installed).

    Append-only: every flush (``write_wave``) adds ONE new part per table, ``<table>.w<k>.r<rank>.<ext>`` -- a long table
    (the predictions of a wave) as up to four row pieces ``<table>.w<k>.r<rank>.p<j>.<ext>`` written by as many threads
    (pyarrow releases the GIL; the parquet writer itself is single-threaded) and read back in order -- and
    commits the wave by writing the marker ``_wave.w<k>.r<rank>.ok`` last (files are written to a temporary name and
    renamed, so a part is either complete or absent).  Parts without their marker -- a run killed mid-flush -- are
    ignored by readers and removed by the next run.  Nothing already written is ever re-read or re-written by an
    append (the reference's HDFStore.append, local_experts.py:526-548).  ``put`` writes a whole table
    (``<table>.<ext>``, the HDFStore.put(append=False) of the smoothing step)."""

    def __init__(self, path: Optional[str], rank: int = 0, fmt: Optional[str] = None):
        self.path = path
        self.rank = int(rank)
        if fmt is None:
            fmt = "parquet" if _have_pyarrow() else "pickle"
        if fmt not in ("parquet", "pickle"):
            raise ValueError("fmt must be 'parquet' or 'pickle'")
        if fmt == "parquet" and not _have_pyarrow():
            raise ImportError("fmt='parquet' needs pyarrow")
        self.fmt = fmt
        self.ext = "parquet" if fmt == "parquet" else "pkl"
        self._open_k = None
        if path:
            os.makedirs(path, exist_ok=True)

    def _whole(self, table):
        """Path of the whole-table file of ``table`` (either format), or None."""
        for ext in ("parquet", "pkl"):
            f = os.path.join(self.path, f"{table}.{ext}")
            if os.path.exists(f):
                return f
        return None

    def _scan(self):
        parts, marks = {}, set()
        for f in os.listdir(self.path):
            if f.startswith(".tmp."):                       # unfinished temporary of some flush: never a table part
                continue
            m = _MARK_RE.match(f)
            if m:
                marks.add((int(m["k"]), int(m["r"])))
                continue
            m = _PART_RE.match(f)
            if m:
                parts.setdefault(m["table"], []).append((int(m["k"]), int(m["r"]), f))
        return parts, marks

    def drop_uncommitted(self):
        """Remove part files of this rank that no marker commits (left by a run that died during a flush), and this rank's
        stale temporaries."""
        if not self.path:
            return
        parts, marks = self._scan()
        for plist in parts.values():
            for k, r, f in plist:
                if r == self.rank and (k, r) not in marks:
                    os.remove(os.path.join(self.path, f))
        # temporaries carry their writer's RANK (`.tmp.r<rank>.<pid>.<name>`): a rank only ever removes its own -- ranks that
        # share the directory from different PID namespaces or hosts (one container per rank, NFS) cannot see each other's
        # PIDs, and a late starter must not delete another rank's file in flight.  Temporaries of older versions
        # (`.tmp.<pid>.<name>`) are removed once they are an hour old.
        mine = f".tmp.r{self.rank}."
        for f in os.listdir(self.path):
            stale = False
            if f.startswith(mine):
                pid = f[len(mine):].split(".")[0]
                stale = not (pid.isdigit() and int(pid) == os.getpid())
            elif f.startswith(".tmp.") and not f.startswith(".tmp.r"):
                try:
                    stale = time.time() - os.path.getmtime(os.path.join(self.path, f)) > 3600.0
                except OSError:
                    stale = False
            if stale:
                try:
                    os.remove(os.path.join(self.path, f))
                except FileNotFoundError:
                    pass

    def _atomic_write(self, df, name):
        tmp = os.path.join(self.path, f".tmp.r{self.rank}.{os.getpid()}.{name}")
        if name.endswith(".parquet"):
            # pyarrow directly, without dictionary encoding and column statistics: a part is written once and read whole, and
            # the encoder's dictionaries and min / max passes were four fifths of pandas' to_parquet on the preds table
            # (0.149 -> 0.032 s for 210 k rows; the file is also 13 % smaller)
            import pyarrow as pa
            import pyarrow.parquet as pq
            pq.write_table(pa.Table.from_pandas(df, preserve_index=True), tmp, compression="snappy", use_dictionary=False,
                           write_statistics=False)
        else:
            df.to_pickle(tmp)
        os.replace(tmp, os.path.join(self.path, name))

    def _wave_number(self):
        """The number of the wave being written (fixed by its first piece or, without pieces, by ``write_wave``)."""
        if self._open_k is None:
            _, marks = self._scan()
            self._open_k = 1 + max([kk for kk, r in marks if r == self.rank], default=0)
        return self._open_k

    def write_piece(self, name: str, j: int, df: pd.DataFrame):
        """Row piece ``j`` (0, 1, ...; in row order) of table ``name`` of the wave that the next ``write_wave`` commits: the
        rows of a long table that are ready before the wave is (the predictions of an engine call), written while the wave
        still runs.  Uncommitted like any part until the marker is there."""
        if not self.path or df is None or not len(df):
            return
        assert 0 <= j < 100
        self._atomic_write(df, f"{name}.w{self._wave_number():06d}.r{self.rank:03d}.p{j:02d}.{self.ext}")

    def write_wave(self, tables: Dict[str, pd.DataFrame], prewritten=None):
        """One committed part per non-empty table (``prewritten``: {table: rows} of the tables whose rows went out as
        ``write_piece`` pieces -- they are committed by the same marker)."""
        if not self.path:
            return
        prewritten = dict(prewritten or {})
        tables = {k: v for k, v in tables.items() if v is not None and len(v)}
        if not tables and not prewritten:
            self._open_k = None
            return
        k = self._wave_number()
        self._open_k = None
        jobs = []
        for name, df in tables.items():
            if name in prewritten:
                continue
            stem = f"{name}.w{k:06d}.r{self.rank:03d}"
            npiece = min(_PIECE_MAX, -(-len(df) // _PIECE_ROWS)) if self.fmt == "parquet" else 1
            if npiece <= 1:
                jobs.append((df, f"{stem}.{self.ext}"))
            else:
                cut = np.linspace(0, len(df), npiece + 1).astype(np.int64)
                jobs += [(df.iloc[cut[j]:cut[j + 1]], f"{stem}.p{j:02d}.{self.ext}") for j in range(npiece)]
        if len(jobs) == 1:
            self._atomic_write(*jobs[0])
        elif jobs:
            for f_ in [_writer_pool().submit(self._atomic_write, d_, n_) for d_, n_ in jobs]:
                f_.result()
        mark = os.path.join(self.path, f"_wave.w{k:06d}.r{self.rank:03d}.ok")
        with open(mark + ".tmp", "w") as f:
            rows = {n: int(len(d)) for n, d in tables.items()}
            rows.update({n: int(r) for n, r in prewritten.items()})
            f.write(json.dumps({"tables": sorted(rows), "rows": rows}))
        os.replace(mark + ".tmp", mark)

    def append(self, table, df: pd.DataFrame):
        self.write_wave({table: df})

    def put(self, table, df: pd.DataFrame):
        """Whole-table write: replaces the table and every part of it."""
        if not self.path:
            return
        parts, _ = self._scan()
        for _, _, f in parts.get(table, []):
            os.remove(os.path.join(self.path, f))
        old = self._whole(table)
        self._atomic_write(df, f"{table}.{self.ext}")
        if old is not None and not old.endswith("." + self.ext):
            os.remove(old)

    def read(self, table) -> Optional[pd.DataFrame]:
        if not self.path:
            return None
        parts, marks = self._scan()
        pieces = []
        whole = self._whole(table)
        if whole is not None:
            pieces.append(_read_part(whole))
        for k, r, f in sorted(parts.get(table, [])):
            if (k, r) in marks:
                pieces.append(_read_part(os.path.join(self.path, f)))
        if not pieces:
            return None
        return pieces[0] if len(pieces) == 1 else pd.concat(pieces)

    def table_names(self) -> List[str]:
        if not self.path:
            return []
        parts, marks = self._scan()
        names = {t for t, pl in parts.items() if any((k, r) in marks for k, r, _ in pl)}
        for f in os.listdir(self.path):
            if f.startswith(".tmp.") or _PART_RE.match(f):
                continue
            m = _WHOLE_RE.match(f)
            if m:
                names.add(m["table"])
        return sorted(names)

    def tables(self) -> Dict[str, pd.DataFrame]:
        return {t: self.read(t) for t in self.table_names()}


def export_parquet(store_path: str, out_dir: str, expert_order: bool = True) -> List[str]:
    """ONE parquet file per table (``<out_dir>/<table>.parquet``, rows in expert order), from a store of either format:
    what a consumer outside this package reads with ``pandas.read_parquet``.  Counterpart of handing the reference's
    HDF5 results file to downstream tooling (``get_results_from_h5file``, GPSat/local_experts.py:1467-1620)."""
    if not _have_pyarrow():
        raise ImportError("export_parquet needs pyarrow")
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for name, df in get_results(store_path, expert_order=expert_order).items():
        if df is None:
            continue
        f = os.path.join(out_dir, f"{name}.parquet")
        df.to_parquet(f, engine="pyarrow", index=True)
        written.append(f)
    cfgs = [f for f in os.listdir(store_path) if f.startswith("oi_config") and f.endswith(".json")]
    for f in cfgs:
        with open(os.path.join(store_path, f)) as src, open(os.path.join(out_dir, f), "w") as dst:
            dst.write(src.read())
    return written


def export_hdf5(store_path: str, h5_path: str, expert_order: bool = True):
    """The store as ONE HDF5 file in the reference's layout (``pd.HDFStore`` tables keyed by table name, appendable
    format with data columns), for GPSat's own readers (``get_results_from_h5file``, GPSat/local_experts.py:1467).  Needs
    pytables, which is not a dependency of this backend: raises ImportError when it is absent (it is absent from the
    build and GPU images, so this path has not been executed there)."""
    try:
        import tables  # noqa: F401
    except Exception as e:                                   # pragma: no cover
        raise ImportError("export_hdf5 needs pytables (pip install tables)") from e
    with pd.HDFStore(h5_path, mode="w") as st:               # pragma: no cover
        for name, df in get_results(store_path, expert_order=expert_order).items():
            if df is not None and len(df):
                st.put(name, df, format="table", data_columns=True)


def get_results(store_path: str, expert_order: bool = False) -> Dict[str, pd.DataFrame]:
    """Counterpart of ``get_results_from_h5file`` (GPSat/local_experts.py:1467): all tables of a store.  With
    ``expert_order=True`` the rows of every table are put in the order of the ``expert_locs`` table (a sharded run
    commits its parts in (wave, rank) order)."""
    tabs = ResultStore(store_path).tables()
    if expert_order:
        xl = next((v for k, v in tabs.items() if k.startswith("expert_locs")), None)
        if xl is not None:
            for k, v in tabs.items():
                if k.startswith("expert_locs") or v.index.names != xl.index.names or len(v) == 0:
                    continue
                pos = xl.index.get_indexer(v.index)
                tabs[k] = v.iloc[np.argsort(pos, kind="stable")]
    return tabs


def check_prev_oi_config(prev_oi_config: dict, oi_config: dict, skip_valid_checks_on=None):
    """What ``GPSat.utils.check_prev_oi_config`` (utils.py:1276-1327) is documented to do: raise when a key of the
    current configuration, other than those in ``skip_valid_checks_on``, differs from the one a previous run stored
    for the same tables.  (The reference's final ``assert len(bad_keys)`` tests the opposite of its docstring.)"""
    skip = set(skip_valid_checks_on or [])
    bad = [k for k, v in oi_config.items() if k not in skip and prev_oi_config.get(k) != v]
    assert not bad, f"the following keys did not have values that matched exactly: {bad}"


def _adjust_func(spec):
    f = spec.get("func") if isinstance(spec, dict) else spec
    if callable(f):
        return f
    if isinstance(f, str) and f.lstrip().startswith("lambda"):
        return eval(f)                                     # the reference's config_func does the same (utils.py:311)
    raise NotImplementedError("load_params.index_adjust takes {col: {'func': callable or 'lambda x: ...'}}")


# ----------------------------------------------------------------------------------------------------------
# the batched orchestrator
# ----------------------------------------------------------------------------------------------------------
class BatchedLocalExpertOI:
    def __init__(self, expert_loc_config: dict, data_config: dict, model_config: dict, pred_loc_config: dict,
                 engine=None, device_select: bool = False, dtype: str = "f32"):
        self.config = {"locations": _jsonable(expert_loc_config), "data": _jsonable(data_config),
                       "model": _jsonable(model_config), "pred_loc": _jsonable(pred_loc_config)}
        if dtype not in DTYPES:
            raise ValueError("dtype must be 'f32' (default) or 'f64' (the reference's precision)")
        self.dtype = dtype
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
        self.params_to_store = model_config.get("params_to_store") or PARAM_NAMES
        self.load_params = model_config.get("load_params")
        # load_params.previous (local_experts.py:1059-1064): start every tile from the running average of earlier optima.
        # What the reference's load_params does with the combinations (local_experts.py:553-609: `if file is not None: ...
        # elif previous is not None: param_dict = previous_params`):
        #   file + previous            the FILE's parameters are set, the running average is carried along and never used;
        #   previous + direct values   the direct values are never applied (previous=True: the running average;
        #                              previous=False: nothing is set at all);
        # and `previous` among the keys makes _same_param_table False (:749-758), i.e. parameters are always stored.
        self.use_previous = False
        self._lp_keys = set(self.load_params) if self.load_params is not None else set()
        lp = self.load_params
        if lp is not None and lp.get("file") is None and lp.get("previous") is not None:
            self.use_previous = bool(lp["previous"])
            self.load_params = None
        elif lp is not None and "previous" in lp:
            self.load_params = {k: v for k, v in lp.items() if k not in ("previous", "previous_params")}
        # ---- prediction locations (local_experts.py:254-264)
        plc = dict(pred_loc_config or {"method": "expert_loc"})
        self.pred_loc = PredictionLocations(coords_col=self.coords_col, **plc)
        from .engine import default_engine
        self.engine = engine if engine is not None else default_engine()
        self.engine_workers = 2          # engines (HIP streams) that take the chunks of a wave in turn; see run_shard
        self._extra_engines = []
        self.pack_threads = 4            # host threads that pack one engine call's arrays
        self._pack_pool = None
        # tile membership for all experts in one GPU call (bit-identical to the host selector)
        self.device_select = device_select
        self.timings = {}

    # ------------------------------------------------------------------------------------------------------
    # per-profile template: everything HipGPRModel's constructor + set_parameter_constraints decide that does not
    # depend on the tile's rows (defaults, scales, box, trainable mask).  One throw-away model on a two-row frame,
    # as the reference itself does to read param_names (postprocessing.py:202-211).
    # ------------------------------------------------------------------------------------------------------
    def _template(self, pf):
        D = len(self.coords_col)
        dummy = pd.DataFrame({**{c: [0.0, 1.0] for c in self.coords_col}, self.obs_col: [0.0, 1.0]})
        m = HipGPRModel(data=dummy, obs_col=self.obs_col, coords_col=self.coords_col, engine=self.engine,
                        verbose=False, dtype=self.dtype, **pf["init_params"])
        theta_default = m._theta.copy()
        cons = None
        if pf["constraints"] is not None:
            cons = {k: dict(v) for k, v in pf["constraints"].items()}
            if pf["init_params"].get("coords_scale", None) is not None and "lengthscales" in cons:
                cons["lengthscales"]["scale"] = True           # local_experts.py:1113-1114
            m.set_parameter_constraints(cons, move_within_tol=True, tol=1e-2)
        ok = pf["optim_kwargs"]
        m._fix_hyperparameters(list(ok.get("fixed_params") or []))
        # per constrained slice: the tolerance the clamp uses (gpflow_models.py:471-479 with run()'s tol = 1e-2)
        clamp = []
        for pn, c in (cons or {}).items():
            sl = m._slice(pn)
            if c.get("move_within_tol", True):
                clamp.append((sl, float(c.get("tol", 1e-2))))
        return dict(theta_default=theta_default, lo=m._lo.copy(), hi=m._hi.copy(), trainable=m._trainable.copy(),
                    clamp=clamp, coords_scale=np.broadcast_to(m.coords_scale, (1, D)).astype(np.float64),
                    obs_scale=float(m.obs_scale.reshape(-1)[0]),
                    local_mean=isinstance(pf["init_params"].get("obs_mean"), str) and pf["init_params"]["obs_mean"] == "local",
                    unconstrained_noise=not np.isfinite(m._lo[D + 1]), device=str(m.gpu_name)[:64])

    def _load_param_tables(self, store: "ResultStore", table_suffix):
        """load_params tables, read ONCE per run and indexed by expert coordinates (local_experts.py:553-689)."""
        lp = self.load_params
        src = lp.get("file")
        tsuf = lp.get("table_suffix", "")                  # load_params(table_suffix="") default, local_experts.py:561
        names = lp.get("param_names") or PARAM_NAMES
        out = {}
        reader = None if isinstance(src, dict) else ResultStore(src)
        for pn in names:
            assert pn in PARAM_NAMES, f"provide param name:{pn}\nis not in param_names:{PARAM_NAMES}"
            tab = src.get(f"{pn}{tsuf}") if isinstance(src, dict) else reader.read(f"{pn}{tsuf}")
            if tab is not None and len(tab):
                out[pn] = tab
        return out

    def _loaded_theta(self, tabs, locs, theta, unconstrained_noise):
        """Overwrite rows of ``theta`` [T, H] with the stored parameters of each expert location.  Returns the mask of
        tiles for which at least one parameter was found (the others are skipped, local_experts.py:1099-1101)."""
        cc, D = self.coords_col, len(self.coords_col)
        found_any = np.zeros(len(locs), dtype=bool)
        look = locs.copy()
        for col, spec in (self.load_params.get("index_adjust") or {}).items():
            j = cc.index(col)
            f = _adjust_func(spec)
            look[:, j] = [f(v) for v in look[:, j]]
        key = _index_for(cc, look)
        slots = {"lengthscales": (0, D), "kernel_variance": (D, 1), "likelihood_variance": (D + 1, 1)}
        for pn, tab in tabs.items():
            start, width = slots[pn]
            colvals = np.full((len(locs), width), np.nan)
            for k in range(width):
                sub = tab[tab["_dim_0"] == k] if "_dim_0" in tab.columns else tab
                sub = sub[~sub.index.duplicated(keep="first")]
                pos = sub.index.get_indexer(key)
                col = sub[pn].values.astype(np.float64)
                colvals[:, k] = np.where(pos >= 0, col[np.clip(pos, 0, max(len(col) - 1, 0))], np.nan) if len(col) else np.nan
            good = ~np.isnan(colvals).any(axis=1)             # NaN -> parameter dropped (local_experts.py:670-679)
            if pn == "likelihood_variance" and unconstrained_noise:
                low = good & (colvals[:, 0] < LIKELIHOOD_VARIANCE_LOWER_BOUND)
                if low.any():
                    warnings.warn("likelihood_variance below variance_lower_bound: set to the bound (gpflow_models.py:404-409)")
                colvals[low, 0] = LIKELIHOOD_VARIANCE_LOWER_BOUND
            theta[good, start:start + width] = colvals[good]
            found_any |= good
        return found_any

    # ------------------------------------------------------------------------------------------------------
    def run(self, store_path: Optional[str] = None, optimise: bool = True, predict: bool = True, min_obs: int = 3,
            table_suffix: str = "", max_tiles_per_call: Optional[int] = None, store_every: Optional[int] = None,
            check_config_compatible: bool = True, skip_valid_checks_on: Optional[List[str]] = None,
            rank: Optional[int] = None, world_size: Optional[int] = None, gather: bool = True,
            engine_chunk: Optional[int] = None):
        """See the module docstring.  ``store_every``: expert locations per flushed wave (default 4096;
        ``max_tiles_per_call`` is the older name of the same knob).  ``engine_chunk``: tiles per engine call inside a wave
        (default 1024: with two engines the kernel of one call runs while the other call's arrays are copied and unpacked, and the
        tail of one kernel is filled by the next -- 4096 experts 18.0 -> 19.3 k tiles/s, 16 384 experts 19.5 -> 22.2 k against calls of
        4096): while the GPU works on one call the host packs the next (gather, scale, de-mean, centre, cast).  ``rank`` / ``world_size``: tile-sharded run, one
        process per GPU (default: taken from an initialised ``torch.distributed`` group, else 0 / 1); with
        ``gather=True`` rank 0 returns the global tables in expert order, the other ranks their own shard's
        (``gather="always"`` runs the exchange in a group of one rank too).
        ``world_size > 1`` with ``rank=None`` and no process group runs all the LOGICAL shards one after the other on
        this process's engine and merges them with the routine that closes the gather -- the one-GPU rehearsal of the
        multi-GPU path."""
        t_start = time.perf_counter()
        d_rank, d_world = _dist_rank_world()
        logical = world_size is not None and world_size > 1 and rank is None and d_world == 1
        if logical:
            rank = 0
        elif rank is None or world_size is None:
            rank, world_size = d_rank, d_world
        in_group = (not logical) and world_size > 1 and d_world == world_size
        if world_size > 1 and not logical and not in_group:
            # explicit rank / world_size without a process group of that size: the caller synchronises the ranks itself
            # (nothing here can separate reading the resume state from rank 0's writes, and nothing can gather)
            if gather:
                raise RuntimeError(f"run(rank={rank}, world_size={world_size}, gather=True) needs an initialised "
                                   f"torch.distributed group of {world_size} ranks (found {d_world}); pass gather=False to run "
                                   f"this rank's shard on its own (start the ranks only after the store directory exists)")
        store = ResultStore(store_path, rank=rank)
        cc = self.coords_col
        D, H = len(cc), len(cc) + 2
        xl = self.expert_locs
        wave_n = int(store_every or max_tiles_per_call or 4096)
        chunk_n = max(1, int(engine_chunk or 1024))
        # ---- expert_locs table + config bookkeeping (local_experts.py:873-903); rank 0 owns the shared files
        config_id = 1
        if store_path:
            store.drop_uncommitted()
            cfg_file = os.path.join(store_path, f"oi_config{table_suffix}.json")
            prev = json.load(open(cfg_file)) if os.path.exists(cfg_file) else []
            if prev and check_config_compatible:
                check_prev_oi_config(prev[-1]["config"], self.config, skip_valid_checks_on)
            config_id = len(prev) + 1
            done = store.read(f"run_details{table_suffix}")          # resume state, read before any rank writes
            if in_group:
                import torch.distributed as dist
                dist.barrier()
            if rank == 0:
                prev.append({"idx": config_id, "datetime": time.strftime("%Y-%m-%d %H:%M:%S"), "config": self.config,
                             "run_kwargs": {"optimise": optimise, "predict": predict, "min_obs": min_obs,
                                            "table_suffix": table_suffix, "store_every": wave_n, "dtype": self.dtype}})
                with open(cfg_file + ".tmp", "w") as f:
                    json.dump(prev, f)
                os.replace(cfg_file + ".tmp", cfg_file)
                if store.read(f"expert_locs{table_suffix}") is None:
                    store.put(f"expert_locs{table_suffix}", xl.set_index(cc))
        # ---- resume: drop expert locations already in run_details (local_experts.py:475-497,908-912)
        todo = np.ones(len(xl), dtype=bool)
        if not store_path:
            done = None
        if done is not None and len(done):
            todo = ~np.asarray(_index_for(cc, xl[cc].values.astype(np.float64)).isin(done.index))
        ex = np.nonzero(todo)[0]                                   # global expert positions still to run
        locs = xl[cc].values.astype(np.float64)[ex]

        # ---------------- pass 1: membership, prediction coordinates, parameter vectors (whole-array) ----------------
        self.timings["setup_s"] = time.perf_counter() - t_start
        self.timings["flush_wait_s"] = 0.0
        t0 = time.perf_counter()
        refs = xl.iloc[ex]
        # the prediction locations of all experts on a second engine (another HIP stream) while the first selects the observations
        pcs_f = None
        sel_engines = self._engine_pool(2) if (self.device_select and len(ex)) else [self.engine]
        if len(sel_engines) > 1:
            from concurrent.futures import ThreadPoolExecutor
            sel_pool = ThreadPoolExecutor(max_workers=1)
            pcs_f = sel_pool.submit(self.pred_loc.batch, locs, sel_engines[1])
        if len(self.local_select):
            sel = DeviceSelector(self.df, self.local_select, self.engine) if self.device_select \
                else LocalSelector(self.df, self.local_select)
            off, idx = sel.select(refs)
        else:
            off, idx = np.arange(len(ex) + 1, dtype=np.int64) * len(self.df), np.tile(np.arange(len(self.df)), len(ex))
        n_obs = np.diff(off)
        if pcs_f is not None:
            pcs = pcs_f.result()
            sel_pool.shutdown(wait=True)
        else:
            pcs = self.pred_loc.batch(locs, self.engine if self.device_select else None) if len(ex) \
                else RaggedRows(np.zeros((0, D)), np.zeros(1, dtype=np.int64))
        n_pred = pcs.counts.astype(np.int64)
        self.timings["select_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        # item kinds: 0 skipped silently (no prediction locations, local_experts.py:962-965), 1 stub row
        # (N < min_obs, :988-1012), 2 tile, 3 error row (tile larger than the kernels take)
        kind = np.full(len(ex), 2, dtype=np.int8)
        # largest tile the kernels take (gpsat_max_tile_obs): larger ones get an explicit error row instead of failing
        # the whole batch
        max_obs = L.max_tile_obs(self.dtype, D)
        kind[n_obs > max_obs] = 3
        kind[n_obs < min_obs] = 1
        kind[n_pred == 0] = 0
        if (kind == 3).any():
            warnings.warn(f"{int((kind == 3).sum())} expert locations select more than {max_obs} "
                          f"observations: not run (error row in run_details)")
        is_repl = (n_obs < self.replacement_threshold) if self.replacement_threshold is not None \
            else np.zeros(len(ex), dtype=bool)                      # local_experts.py:1021-1041
        prof_names = list(self.profiles)
        prof_id = np.where(is_repl, prof_names.index("replacement") if "replacement" in prof_names else 0, 0)
        tmpl, pinfo = {}, {}
        for pi, pname in enumerate(prof_names):
            pf = self.profiles[pname]
            kernel = pf["init_params"].get("kernel", "Matern32")
            if kernel not in L.KERNEL_IDS:
                raise NotImplementedError(f"kernel {kernel!r}")
            ok = pf["optim_kwargs"]
            tmpl[pi] = self._template(pf)
            pinfo[pi] = dict(kernel=kernel, max_iter=int(ok.get("max_iter", 10_000)),
                             eng_kw={k: ok[k] for k in ("max_ls", "ftol", "gtol", "adam_lr") if k in ok},
                             optimiser=ok.get("optimiser", "lbfgs") if optimise else "none",
                             apply_scale=pf["pred_kwargs"].get("apply_scale", True),
                             full_cov=bool(pf["pred_kwargs"].get("full_cov", False)) and predict)
        theta0 = np.zeros((len(ex), H))
        lo = np.full((len(ex), H), np.nan)
        hi = np.full((len(ex), H), np.nan)
        save_params = np.ones(len(ex), dtype=bool)
        for pi in tmpl:
            m_ = prof_id == pi
            theta0[m_], lo[m_], hi[m_] = tmpl[pi]["theta_default"], tmpl[pi]["lo"], tmpl[pi]["hi"]
        if self.load_params is not None:
            lp = self.load_params
            if lp.get("file") is not None:
                tabs = self._load_param_tables(store, table_suffix)
                for pi in tmpl:
                    m_ = np.nonzero((prof_id == pi) & (kind == 2))[0]
                    th = theta0[m_]
                    got = self._loaded_theta(tabs, locs[m_], th, tmpl[pi]["unconstrained_noise"])
                    theta0[m_] = th
                    kind[m_[~got]] = 0                              # nothing loadable: tile skipped (:1099-1101)
                same = (lp.get("file") == store_path and lp.get("table_suffix", None) == table_suffix and
                        self._lp_keys <= {"file", "table_suffix"})           # _same_param_table, :749-758
                save_params[:] = not (same and not optimise)        # local_experts.py:1090-1097
            else:
                # parameters given directly (load_params(**param_dict), local_experts.py:553-604)
                direct = {k: v for k, v in lp.items() if k in PARAM_NAMES}
                if not direct:
                    raise NotImplementedError("load_params needs 'file' or parameter values")
                for pn, v in direct.items():
                    v = np.asarray(v, dtype=np.float64).reshape(-1)
                    if pn == "lengthscales":
                        theta0[:, :D] = v
                    else:
                        theta0[:, D + PARAM_NAMES.index(pn) - 1] = v[0]
        for pi, t_ in tmpl.items():                                  # move within tol of the box (gpflow_models.py:471-479)
            m_ = prof_id == pi
            for sl, tol in t_["clamp"]:
                theta0[m_, sl] = clamp_within(theta0[m_, sl], t_["lo"][sl], t_["hi"][sl], tol)
        self.timings["params_s"] = time.perf_counter() - t0
        want_cov = any(p_["full_cov"] for p_ in pinfo.values())

        # ---------------- shard: LPT on the cost model over the items of this run ----------------
        cost_n = np.where(kind == 2, n_obs, 0)
        if world_size > 1:
            parts = sharding.partition_tiles(cost_n, np.where(kind == 2, n_pred if predict else 0, 0), world_size)
        else:
            parts = [np.arange(len(ex), dtype=np.int64)]
        parts = [p_[kind[p_] != 0] for p_ in parts]                  # silently skipped locations produce nothing

        coords_all = self.df.loc[:, cc].values.astype(np.float64)
        obs_all = self.df[self.obs_col].values.astype(np.float64)
        assert not np.isnan(coords_all).any(), "nans found in coords"
        assert not np.isnan(obs_all).any(), "nans found in obs"
        self.timings.update(engine_s=0.0, engine_call_s=0.0, kernel_s=0.0, tables_s=0.0, flush_s=0.0)
        self.timings["calls"] = []          # per engine call: (job, tiles, start, end, kernel seconds), times from the start of run()

        def tables_for(items, fixed, pred_cat, cov_cat=None, with_preds=True):
            return self._tables(ex[items], locs[items], kind[items], n_obs[items], fixed, pred_cat,
                                (pcs, items) if predict else None, save_params[items],
                                [tmpl[p]["device"] for p in prof_id[items]], optimise, config_id, table_suffix,
                                cov_cat=cov_cat, cov_tiles=(kind[items] == 2) & np.array([pinfo[p]["full_cov"] for p in prof_id[items]], dtype=bool)
                                if want_cov else None, with_preds=with_preds)

        # ---------------- pass 2: waves of one shard ----------------
        def pack_job(ids, pi):
            """Host-side intake of one engine call (a1 of the reference in fp64): gather the tiles' rows, scale, de-mean, centre,
            cast -- by `pack_threads` threads, each writing its range of tiles into the call's arrays (NumPy releases the GIL
            in the gathers and the arithmetic)."""
            t_, p_ = tmpl[pi], pinfo[pi]
            Ns = n_obs[ids]
            o_off = np.concatenate([[0], np.cumsum(Ns)]).astype(np.int64)
            Ps = n_pred[ids] if predict else np.zeros(len(ids), dtype=np.int64)
            p_off = np.concatenate([[0], np.cumsum(Ps)]).astype(np.int64)
            out_dt = np.float32 if self.dtype == "f32" else np.float64
            X = np.empty((int(o_off[-1]), D), dtype=out_dt)
            y = np.empty(int(o_off[-1]), dtype=out_dt)
            Xs = np.empty((int(p_off[-1]), D), dtype=out_dt)
            mean = np.zeros(len(ids))
            consecutive = len(ids) > 0 and bool(np.all(np.diff(ids) == 1))

            def sub(a, b):
                sl = ids[a:b]
                if consecutive:
                    rows = idx[off[sl[0]]:off[sl[-1] + 1]]
                else:
                    rows = np.concatenate([idx[off[i]:off[i + 1]] for i in sl])
                ns, oo = Ns[a:b], o_off[a:b + 1] - o_off[a]
                Xd = coords_all[rows] / t_["coords_scale"]        # base_model.py:243
                yv_ = obs_all[rows]
                if t_["local_mean"]:
                    mean[a:b] = np.add.reduceat(yv_, oo[:-1]) / ns
                yd = (yv_ - np.repeat(mean[a:b], ns)) / t_["obs_scale"]   # base_model.py:244-245
                if predict:
                    Xsd = pcs.take(sl)
                    if p_["apply_scale"]:
                        Xsd = Xsd / t_["coords_scale"]
                else:
                    Xsd = np.zeros((0, D))
                if self.dtype == "f32" and len(Xd):
                    # what the engine does with fp64 host arrays for the fp32 kernels (per-tile centring, then the cast), done
                    # here so that it too overlaps the previous call
                    from .engine import centre_tiles
                    Xd, Xsd = centre_tiles(Xd, Xsd, oo, p_off[a:b + 1] - p_off[a])
                X[o_off[a]:o_off[b]] = Xd
                y[o_off[a]:o_off[b]] = yd
                Xs[p_off[a]:p_off[b]] = Xsd

            nsub = max(1, min(self.pack_threads, len(ids) // 128))
            bounds = np.linspace(0, len(ids), nsub + 1).astype(np.int64)
            if nsub == 1:
                sub(0, len(ids))
            else:
                for f_ in [self._sub_pool().submit(sub, int(bounds[j]), int(bounds[j + 1])) for j in range(nsub)]:
                    f_.result()
            return dict(o_off=o_off, X=X, y=y, p_off=p_off, Xs=Xs, mean=mean)

        # ---------------- pass 2: waves of one shard ----------------
        # A wave (the flush unit) is cut into engine calls of at most `engine_chunk` tiles; the next call's arrays are packed
        # by a helper thread while the GPU works on the current one (ctypes releases the GIL during the call).  Per-tile
        # results do not depend on how tiles are batched (tests/test_gpu_parity.py::test_ragged_batch_tile_indexing_is_bit_exact).
        def run_shard(mine, shard_store):
            from concurrent.futures import ThreadPoolExecutor
            fixed_rows, pred_rows, cov_rows = [], [], []
            # load_params.previous: the running average of earlier optima (rho = 0.95, local_experts.py:1200-1217); it starts
            # from the default parameters of the first model built (:1053-1054)
            prev = {"theta": None}
            jobs = []                                                  # (wave index, profile, positions within the wave)
            waves = [mine[w0:w0 + wave_n] for w0 in range(0, len(mine), wave_n)]
            for wi, items in enumerate(waves):
                for pi in tmpl:
                    loc_ids = np.nonzero((kind[items] == 2) & (prof_id[items] == pi))[0]
                    for c0 in range(0, len(loc_ids), chunk_n):
                        jobs.append((wi, pi, loc_ids[c0:c0 + chunk_n]))
            last_job_of_wave = {wi: k for k, (wi, _, _) in enumerate(jobs)}
            state = {}
            # The predictions are most of a wave's bytes: when a wave's tiles run as consecutive calls of ONE model profile (the
            # rows of its preds table are then the calls' rows back to back), every call's rows go to the store as a row piece
            # as soon as the call returns, and the wave's flush is left with the small tables and the marker.
            jobs_of_wave = {}
            for k, (wi, pi, _) in enumerate(jobs):
                jobs_of_wave.setdefault(wi, []).append(k)
            incremental = {wi: bool(shard_store.path) and predict and len(ks) <= 64 and len({jobs[k][1] for k in ks}) == 1
                           for wi, ks in jobs_of_wave.items()}
            piece_futs = {}
            preds_name = f"preds{table_suffix}"

            def open_wave(wi):
                items = waves[wi]
                state[wi] = (np.full((len(items), H + 6), np.nan),       # theta, nll, status, n_eval, n_iter, seconds, obs mean
                             [np.zeros((0, 3))] * len(items), [np.zeros((0, 2))] * len(items))

            def close_wave(wi):
                items = waves[wi]
                fixed, preds, covs = state.pop(wi)
                tt = time.perf_counter()
                pred_cat = np.concatenate(preds) if len(preds) else np.zeros((0, 3))
                cov_cat = (np.concatenate(covs) if len(covs) else np.zeros((0, 2))) if want_cov else None
                pf_ = piece_futs.pop(wi, [])
                # the predictions went out as pieces: the wave's flush needs the small tables only, and the wave's preds frame
                # is built (after the flush is queued) only where it is returned -- the only wave of a run
                lazy = bool(pf_) and not want_cov
                tables = tables_for(items, fixed, pred_cat, cov_cat, with_preds=not lazy)
                self.timings["tables_s"] += time.perf_counter() - tt
                tf = time.perf_counter()
                # commit: these experts are done.  The parts are written by a helper thread while the next wave runs (one
                # writer, waves in order, marker last); a flush that fails surfaces when the next one is queued or at the end
                if pending:
                    pending.pop().result()

                def commit(pf_=pf_, tables=dict(tables), n_pred_rows=len(pred_cat)):
                    for f_ in pf_:                                     # done by now (one writer, in order): a failed piece fails the wave
                        f_.result()
                    shard_store.write_wave(tables, prewritten={preds_name: n_pred_rows} if pf_ else {})
                pending.append(flusher.submit(commit))
                self.timings["flush_s"] += time.perf_counter() - tf
                if len(waves) == 1:
                    if lazy:
                        tt = time.perf_counter()
                        tile_ = kind[items] == 2
                        cnt_ = np.where(tile_, pcs.counts[items], 0).astype(np.int64)
                        tables[preds_name] = self._preds_frame(locs[items], fixed[:, H + 5], pred_cat, pcs,
                                                               np.asarray(items, dtype=np.int64), tile_, cnt_)
                        self.timings["tables_s"] += time.perf_counter() - tt
                    single["tables"] = tables                          # the only wave's tables ARE the shard's tables
                fixed_rows.append(fixed)
                pred_rows.append(pred_cat)
                if want_cov:
                    cov_rows.append(cov_cat)

            pending = []
            single = {}
            # Engine calls of consecutive chunks are issued from `n_eng` threads, each with an engine (HIP stream, workspace)
            # of its own: the second call's kernel is queued on the GPU while the first one runs and its workgroups take
            # over the CUs the first one's tail leaves idle; packing, the copies and the unpacking of one call overlap the
            # kernel of the other.  `load_params.previous` makes every call depend on the one before: one engine then.
            engines = self._engine_pool(1 if self.use_previous else self.engine_workers)
            n_eng = len(engines)
            import queue
            free_engines = queue.Queue()
            for e_ in engines:
                free_engines.put(e_)
            pk_f, r_f = {}, {}

            def call_engine(k, th_override=None):
                wi, pi, loc_ids = jobs[k]
                ids = waves[wi][loc_ids]
                pk = pk_f.pop(k).result()
                t_, p_ = tmpl[pi], pinfo[pi]
                eng_ = free_engines.get()
                try:
                    te = time.perf_counter()
                    r = eng_.fit_predict_batch(D=D, obs_off=pk["o_off"], X=pk["X"], y=pk["y"], pred_off=pk["p_off"],
                                               Xs=pk["Xs"], theta0=theta0[ids] if th_override is None else th_override,
                                               lo=lo[ids], hi=hi[ids], trainable=t_["trainable"], kernel=p_["kernel"],
                                               optimiser=p_["optimiser"], max_iter=p_["max_iter"],
                                               dtype=self.dtype, **p_["eng_kw"],
                                               **({"full_cov": True} if p_["full_cov"] else {}))
                    t1 = time.perf_counter()
                    self.timings["calls"].append((k, len(ids), round(te - t_start, 4), round(t1 - t_start, 4), round(r.kernel_ms * 1e-3, 4)))
                    return pk, r, t1 - te
                finally:
                    free_engines.put(eng_)

            flusher = ThreadPoolExecutor(max_workers=1)
            with ThreadPoolExecutor(max_workers=max(1, n_eng)) as pack_pool, ThreadPoolExecutor(max_workers=n_eng) as eng_pool:
                try:
                    def submit(k):
                        if k < len(jobs):
                            pk_f[k] = pack_pool.submit(pack_job, waves[jobs[k][0]][jobs[k][2]], jobs[k][1])
                            if not self.use_previous:
                                r_f[k] = eng_pool.submit(call_engine, k)
                    for k in range(min(len(jobs), n_eng + 1)):
                        submit(k)
                    done_waves = 0
                    for k, (wi, pi, loc_ids) in enumerate(jobs):
                        while done_waves < wi:                            # waves without a single model tile (stubs, errors only)
                            if done_waves not in state:
                                open_wave(done_waves)
                            close_wave(done_waves)
                            done_waves += 1
                        if wi not in state:
                            open_wave(wi)
                        fixed, preds, covs = state[wi]
                        ids = waves[wi][loc_ids]
                        t_, p_ = tmpl[pi], pinfo[pi]
                        te = time.perf_counter()
                        if self.use_previous:
                            if prev["theta"] is None:
                                prev["theta"] = t_["theta_default"].copy()
                            th_call = np.tile(prev["theta"], (len(ids), 1))
                            for sl, tol in t_["clamp"]:                   # set_parameters(prev), then the constraints' clamp
                                th_call[:, sl] = clamp_within(th_call[:, sl], t_["lo"][sl], t_["hi"][sl], tol)
                            pk, r, call_s = call_engine(k, th_call)
                            for kk in range(len(ids)):                    # expert order; only successful optimisations, no NaN
                                if p_["optimiser"] != "none" and r.status[kk] == 0 and save_params[ids[kk]] \
                                        and not np.isnan(r.theta[kk]).any():
                                    prev["theta"] = 0.95 * prev["theta"] + 0.05 * r.theta[kk]
                        else:
                            pk, r, call_s = r_f.pop(k).result()
                        submit(k + n_eng + 1)
                        self.timings["engine_s"] += time.perf_counter() - te     # what the main thread waited for this call
                        self.timings["engine_call_s"] += call_s                  # the calls themselves (they overlap)
                        self.timings["kernel_s"] += r.kernel_ms * 1e-3
                        dt = call_s / len(ids)
                        fixed[loc_ids, :H] = r.theta
                        fixed[loc_ids, H] = r.nll
                        fixed[loc_ids, H + 1] = r.status
                        fixed[loc_ids, H + 2] = r.n_eval
                        fixed[loc_ids, H + 3] = r.n_iter if getattr(r, "n_iter", None) is not None else np.nan
                        fixed[loc_ids, H + 4] = dt
                        fixed[loc_ids, H + 5] = pk["mean"]
                        if predict:
                            pr = np.stack([np.asarray(r.f_mean, dtype=np.float64), np.asarray(r.f_var, dtype=np.float64),
                                           np.asarray(r.y_var, dtype=np.float64)], axis=1)
                            p_off = pk["p_off"]
                            for kk, j in enumerate(loc_ids):
                                preds[j] = pr[p_off[kk]:p_off[kk + 1]]
                            if incremental.get(wi):
                                piece = self._preds_frame(locs[ids], pk["mean"], pr, pcs, ids, np.ones(len(ids), dtype=bool),
                                                          pcs.counts[ids].astype(np.int64))
                                piece_futs.setdefault(wi, []).append(
                                    flusher.submit(shard_store.write_piece, preds_name, jobs_of_wave[wi].index(k), piece))
                            if p_["full_cov"]:
                                fc = np.asarray(r.f_cov, dtype=np.float64)
                                for kk, j in enumerate(loc_ids):
                                    P_ = int(p_off[kk + 1] - p_off[kk])
                                    fcov = fc[r.cov_off[kk]:r.cov_off[kk + 1]].reshape(P_, P_)
                                    ycov = fcov.copy()                     # y_cov = f*_cov + diag(y_var - f*_var), gpflow_models.py:250-254
                                    seg = pr[p_off[kk]:p_off[kk + 1]]
                                    ycov[np.arange(P_), np.arange(P_)] += seg[:, 2] - seg[:, 1]
                                    covs[j] = np.stack([fcov.reshape(-1), ycov.reshape(-1)], axis=1)
                        if last_job_of_wave[wi] == k:
                            close_wave(wi)
                            done_waves = wi + 1
                    while done_waves < len(waves):
                        if done_waves not in state:
                            open_wave(done_waves)
                        close_wave(done_waves)
                        done_waves += 1
                except BaseException:
                    # after a fault the flush of the last complete wave is on disk before the fault propagates: a committed
                    # wave survives whatever happens to the next one
                    while pending:
                        pending.pop().result()
                    flusher.shutdown(wait=True)
                    raise
                # the last queued flush goes on while the caller's tables are assembled; run() waits for it before it returns
                # (the flusher pool's exit below would wait as well: the writer thread is handed over instead)
                last_flush.extend(pending)
                flusher_keep.append(flusher)
            fixed_all = np.concatenate(fixed_rows) if fixed_rows else np.zeros((0, H + 6))
            preds_all = np.concatenate(pred_rows) if pred_rows else np.zeros((0, 3))
            cnt = np.where(kind[mine] == 2, n_pred[mine] if predict else 0, 0).astype(np.int64)
            cov_all = (np.concatenate(cov_rows) if cov_rows else np.zeros((0, 2))) if want_cov else None
            shard_tables.append(single.get("tables"))
            return fixed_all, preds_all, cnt, mine, cov_all

        last_flush, flusher_keep, shard_tables = [], [], []

        def finish_flushes():
            tf = time.perf_counter()
            while last_flush:
                last_flush.pop().result()
            for fl in flusher_keep:
                fl.shutdown(wait=True)
            del flusher_keep[:]
            self.timings["flush_wait_s"] += time.perf_counter() - tf

        try:
            all_items = np.nonzero(kind != 0)[0]
            if logical:
                # every logical shard on this engine, merged by the routine that closes the gather
                shards = []
                for r_ in range(world_size):
                    st_r = ResultStore(store_path, rank=r_)
                    st_r.drop_uncommitted()
                    shards.append(run_shard(parts[r_], st_r))
                fixed_g, preds_g, _ = sharding.assemble_global([sh[:4] for sh in shards], len(ex))
                cov_g = None
                if want_cov:
                    cc2 = [np.where(np.array([pinfo[p]["full_cov"] for p in prof_id[sh[3]]], dtype=bool), sh[2] ** 2, 0) for sh in shards]
                    cov_g = sharding.assemble_global([(sh[0], sh[4], c2, sh[3]) for sh, c2 in zip(shards, cc2)], len(ex))[1]
                out = tables_for(all_items, fixed_g[all_items], preds_g, cov_g)
            else:
                mine = parts[rank] if world_size > 1 else parts[0]
                fixed_all, preds_all, cnt, _, cov_all = run_shard(mine, store)
                out = None
                if gather and (world_size > 1 or (gather == "always" and d_world == 1 and _dist_initialised())):
                    # ONE exchange of per-tile results (RCCL over xGMI on the GPU node); tables in expert order on rank 0
                    # (gather="always": also in a group of ONE rank -- the one-GPU rehearsal of the exchange on RCCL)
                    dev_id = getattr(self.engine, "device_id", None)
                    got = sharding.gather_arrays(fixed_all, preds_all, cnt, mine, len(ex), world_size, rank, dev_id)
                    got_c = None
                    if want_cov:                                         # the P x P blocks travel the same way (counts P^2)
                        c2 = np.where(np.array([pinfo[p]["full_cov"] for p in prof_id[mine]], dtype=bool), cnt ** 2, 0)
                        got_c = sharding.gather_arrays(fixed_all, cov_all, c2, mine, len(ex), world_size, rank, dev_id)
                    if rank == 0:
                        fixed_g, preds_g, _ = got
                        out = tables_for(all_items, fixed_g[all_items], preds_g, got_c[1] if got_c is not None else None)
                if out is None:
                    out = shard_tables[-1] if shard_tables and shard_tables[-1] is not None else tables_for(mine, fixed_all, preds_all, cov_all)
        finally:
            finish_flushes()                                       # also after a fault: what was committed is on disk
        self.run_seconds = time.perf_counter() - t_start
        self.timings["total_s"] = self.run_seconds
        return out

    def _sub_pool(self):
        if self._pack_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pack_pool = ThreadPoolExecutor(max_workers=self.pack_threads)
        return self._pack_pool

    def _engine_pool(self, n):
        """The engine given to the constructor plus up to n - 1 more on the same device, created on first use and kept (an
        engine that is not a gpsat_amd Engine -- the CPU tests' stand-ins -- is used alone)."""
        from .engine import Engine
        if n > 1 and isinstance(self.engine, Engine):
            try:
                while len(self._extra_engines) < n - 1:
                    self._extra_engines.append(Engine(self.engine.device_id))
            except Exception as e:                                   # not enough memory for a second workspace, ...
                warnings.warn(f"only {1 + len(self._extra_engines)} engine(s) for the chunks of a wave: {e}")
        return [self.engine] + self._extra_engines[:max(0, n - 1)]

    # ------------------------------------------------------------------------------------------------------
    def _preds_frame(self, locs, f_bar, pred_cat, rag, items, tile, cnt):
        """The ``preds`` table of a run of items (``pred_cat``: the predictions of the tiles among them, back to back)."""
        cc = self.coords_col
        tot = int(cnt.sum())
        assert tot == len(pred_cat), (tot, len(pred_cat))
        raw = rag.take(items[tile]) if tot else np.zeros((0, len(cc)))
        dim0 = np.arange(tot) - np.repeat(np.concatenate([[0], np.cumsum(cnt)])[:-1], cnt)
        pr = {"_dim_0": dim0, "f*": pred_cat[:, 0], "f*_var": pred_cat[:, 1], "y_var": pred_cat[:, 2],
              "f_bar": np.repeat(f_bar, cnt)}
        for ci, c_ in enumerate(cc):
            pr[f"pred_loc_{c_}"] = raw[:, ci]
        return pd.DataFrame(pr, index=_index_for_repeated(cc, locs, cnt))

    def _tables(self, ex_ids, locs, kind, n_obs, fixed, pred_cat, pcs, save_params, devices, optimise, config_id,
                table_suffix, cov_cat=None, cov_tiles=None, with_preds=True):
        """Reference-layout tables for a run of items (rows of ``fixed`` align with the items, ``pred_cat`` holds the
        predictions of the tiles among them back to back).  Pure array assembly (GPSat/local_experts.py:691-747)."""
        cc = self.coords_col
        D, H = len(cc), len(cc) + 2
        n = len(ex_ids)
        tile = kind == 2
        status = fixed[:, H + 1]
        out = {}
        out["run_details"] = pd.DataFrame({
            "_dim_0": np.zeros(n, dtype=np.int64), "num_obs": n_obs.astype(np.int64),
            "run_time": np.where(tile, fixed[:, H + 4], np.nan), "objective_value": np.where(tile, fixed[:, H], np.nan),
            "parameters_optimised": np.full(n, bool(optimise)),
            "optimise_success": tile & bool(optimise) & (status == 0),
            "model": np.full(n, MODEL_NAME, dtype=object),
            "device": np.array([d if t else "" for d, t in zip(devices, tile)], dtype=object),
            "config_id": np.full(n, config_id, dtype=np.int64)}, index=_index_for(cc, locs))
        sp = tile & save_params
        slots = {"lengthscales": (0, D), "kernel_variance": (D, 1), "likelihood_variance": (D + 1, 1)}
        for pn in self.params_to_store:
            start, width = slots[pn]
            vals = fixed[sp, start:start + width].reshape(-1)
            out[pn] = pd.DataFrame({"_dim_0": np.tile(np.arange(width), int(sp.sum())), pn: vals},
                                   index=_index_for_repeated(cc, locs[sp], np.full(int(sp.sum()), width)))
        if pcs is not None and not with_preds:
            pass                                                   # the caller has written the rows as pieces and builds the frame itself
        elif pcs is not None:
            rag, items = pcs                                       # the run's prediction coordinates and these items' positions
            items = np.asarray(items, dtype=np.int64)
            cnt = np.where(tile, rag.counts[items], 0).astype(np.int64)
            out["preds"] = self._preds_frame(locs, fixed[:, H + 5], pred_cat, rag, items, tile, cnt)
            if cov_cat is not None:
                # 2-D arrays of the prediction dict -> table "preds_2" with _dim_0, _dim_1 (row-major), local_experts.py:735-745
                c2 = np.where(cov_tiles, cnt, 0)
                tot2 = int((c2 * c2).sum())
                assert tot2 == len(cov_cat), (tot2, len(cov_cat))
                if tot2:
                    d0 = np.concatenate([np.repeat(np.arange(c), c) for c in c2 if c])
                    d1 = np.concatenate([np.tile(np.arange(c), c) for c in c2 if c])
                    out["preds_2"] = pd.DataFrame({"_dim_0": d0, "_dim_1": d1, "f*_cov": cov_cat[:, 0], "y_cov": cov_cat[:, 1]},
                                                  index=_index_for(cc, np.repeat(locs, c2 * c2, axis=0)))
        else:
            out["preds"] = pd.DataFrame()
        return {f"{k}{table_suffix}": v for k, v in out.items()}


def _dist_initialised():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def _dist_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


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
        if callable(v):
            return getattr(v, "__name__", repr(v))
        return v
    return conv(cfg or {})
