"""Post-processing of local-expert results on the MI355X: hyper-parameter smoothing and gluing of overlapping
predictions.  Host-side mirror of GPSat/postprocessing.py (``smooth_hyperparameters`` :117-375,
``glue_local_predictions`` / ``_1d`` / ``_2d`` :447-577): same argument names and meaning, tables in / tables out; the
arithmetic runs in gpsat_post.hip through the C ABI (``gpsat_smooth_batch`` / ``gpsat_glue_batch``).  No CPU fallback:
without the HIP library these functions raise."""
import copy
import json
import os
import re
from typing import Dict, List, Optional, Union

import numpy as np
import pandas as pd

from .engine import default_engine
from .local_experts import ResultStore

GLUE_MAXVARS = 4


def smooth_hyperparameters(result_file: Union[str, Dict[str, pd.DataFrame]], params_to_smooth: List[str],
                           smooth_config_dict: Dict[str, dict], xy_dims: List[str] = ("x", "y"),
                           coords_col: Optional[List[str]] = None, reference_table_suffix: str = "",
                           table_suffix: str = "_SMOOTHED", output_file: Optional[str] = None,
                           all_params: Optional[List[str]] = None, engine=None,
                           save_config_file: bool = True) -> Dict[str, pd.DataFrame]:
    """Smooth hyper-parameter tables ``<param><reference_table_suffix>`` of a result store (directory path, or a dict of
    tables) with a 2-D Gaussian over ``xy_dims``, one slice per unique combination of the other coordinates and
    ``_dim_*`` columns; values are clipped to ``smooth_config[param]["max"/"min"]`` first; NaN results are dropped;
    parameters not smoothed are copied.  Writes ``<param><reference_table_suffix><table_suffix>`` tables to
    ``output_file`` (a store directory; default: the input store) and returns them
    (GPSat/postprocessing.py:215-343).  With ``save_config_file`` and a store on disk, the configurations recorded by the
    runs that produced the tables (``oi_config<reference_table_suffix>.json``) are re-emitted as the predict-only
    configuration of the next step -- ``run_kwargs.optimise = False``, ``table_suffix`` and ``store_path`` pointing at
    the smoothed tables, ``model.load_params = {"file", "table_suffix"}`` -- in
    ``<store>/oi_config<reference_table_suffix><table_suffix>_predict.json`` (GPSat/postprocessing.py:350-380); its path
    is returned under the key ``"__config_file__"``."""
    eng = engine or default_engine()
    tables = result_file if isinstance(result_file, dict) else ResultStore(result_file).tables()
    if all_params is None:
        all_params = [p for p in ("lengthscales", "kernel_variance", "likelihood_variance")
                      if f"{p}{reference_table_suffix}" in tables]
    missing = [p for p in params_to_smooth if p not in smooth_config_dict]
    if missing:
        raise NotImplementedError(f"parameters {missing} have no entry in smooth_config_dict")
    x_col, y_col = xy_dims
    out = {}
    for param in params_to_smooth:
        name = f"{param}{reference_table_suffix}"
        if name not in tables:
            raise NotImplementedError(f"parameter: {name} is not in tables: {list(tables.keys())}")
        df = tables[name].reset_index()
        cc = coords_col or [c for c in tables[name].index.names if c is not None]
        cfg = smooth_config_dict[param]
        org_cols = df.columns.tolist()
        other_dims = [c for c in cc if c not in xy_dims] + [c for c in df.columns if re.search(r"^_dim_\d", c)]
        pieces = []
        groups = df.groupby(other_dims, sort=False) if other_dims else [((), df)]
        for _, sub in groups:
            sub = sub.copy()
            vals = sub[param].values.astype(np.float64)
            if cfg.get("max") is not None:
                vals[vals > cfg["max"]] = cfg["max"]
            if cfg.get("min") is not None:
                vals[vals < cfg["min"]] = cfg["min"]
            sub[param] = eng.smooth_batch(sub[x_col].values, sub[y_col].values, vals, cfg["l_x"], cfg["l_y"])
            pieces.append(sub.dropna(subset=[param, x_col, y_col])[org_cols])
        out[f"{name}{table_suffix}"] = pd.concat(pieces).set_index(cc)
    for param in all_params:
        if param in params_to_smooth:
            continue
        name = f"{param}{reference_table_suffix}"
        if name in tables:
            out[f"{name}{table_suffix}"] = tables[name].copy(True)
    dest = output_file if output_file is not None else (result_file if isinstance(result_file, str) else None)
    if dest is not None:
        store = ResultStore(dest)
        for k, v in out.items():
            store.put(k, v)                                 # overwrite, like store.put(append=False)
        if save_config_file and isinstance(result_file, str):
            cfg_in = os.path.join(result_file, f"oi_config{reference_table_suffix}.json")
            if os.path.exists(cfg_in):
                new_suffix = f"{reference_table_suffix}{table_suffix}"
                derived = []
                for entry in json.load(open(cfg_in)):
                    oic = copy.deepcopy(entry.get("config", entry))
                    run_kwargs = dict(entry.get("run_kwargs", {}))
                    run_kwargs.update(optimise=False, table_suffix=new_suffix, store_path=dest)
                    model = dict(oic.get("model", {}))
                    model["load_params"] = {"file": dest, "table_suffix": new_suffix}
                    oic["model"] = model
                    oic["run_kwargs"] = run_kwargs
                    derived.append(oic)
                cfg_out = os.path.join(dest, f"oi_config{new_suffix}_predict.json")
                with open(cfg_out, "w") as f:
                    json.dump(derived, f, indent=4)
                out["__config_file__"] = cfg_out
    return out


def _glue(preds_df, pred_loc_cols, xprt_loc_cols, vars_to_glue, inference_radius, R, engine):
    eng = engine or default_engine()
    if isinstance(vars_to_glue, str):
        vars_to_glue = [vars_to_glue]
    if not 1 <= len(vars_to_glue) <= GLUE_MAXVARS:
        raise ValueError(f"1..{GLUE_MAXVARS} variables can be glued per call")
    pred = np.stack([preds_df[c].values.astype(np.float64) for c in pred_loc_cols])
    xprt = np.stack([preds_df[c].values.astype(np.float64) for c in xprt_loc_cols])
    vals = np.stack([preds_df[v].values.astype(np.float64) for v in vars_to_glue])
    # group rows by prediction location: sorted unique keys (what groupby(...).sum() returns), stable order inside
    uniq, inv = np.unique(pred.T, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.argsort(inv, kind="stable")
    seg = np.zeros(len(uniq) + 1, dtype=np.int64)
    np.cumsum(np.bincount(inv, minlength=len(uniq)), out=seg[1:])
    if isinstance(inference_radius, dict):                 # per-expert radius, keyed by expert location (:490-493)
        assert len(xprt_loc_cols) == 1, "a dict of inference radii is a 1-D option in the reference"
        assert len(inference_radius) == len(np.unique(xprt[0]))
        sigma = np.array([inference_radius[loc] for loc in xprt[0]], dtype=np.float64)[order] / R
    elif isinstance(inference_radius, (int, float)):
        sigma = inference_radius / R
    else:
        raise TypeError("inference_radius must be int, float or dict")
    glued = eng.glue_batch(seg, pred[:, order], xprt[:, order], vals[:, order], sigma)
    out = pd.DataFrame({c: uniq[:, i] for i, c in enumerate(pred_loc_cols)})
    for i, v in enumerate(vars_to_glue):
        out[v] = glued[i]
    return out


def glue_local_predictions_1d(preds_df: pd.DataFrame, pred_loc_col: str, xprt_loc_col: str,
                              vars_to_glue: Union[str, List[str]], inference_radius: float, R=3, engine=None):
    """GPSat/postprocessing.py:476-524: one row per unique prediction location, each variable the normal-pdf-weighted
    (std = inference_radius / R, centred on the expert location) average of the overlapping local predictions."""
    return _glue(preds_df, [pred_loc_col], [xprt_loc_col], vars_to_glue, inference_radius, R, engine)


def glue_local_predictions_2d(preds_df: pd.DataFrame, pred_loc_cols: List[str], xprt_loc_cols: List[str],
                              vars_to_glue: Union[str, List[str]], inference_radius: float, R=3, engine=None):
    """GPSat/postprocessing.py:526-577: as 1-D with the product of the two per-axis weights."""
    return _glue(preds_df, list(pred_loc_cols), list(xprt_loc_cols), vars_to_glue, inference_radius, R, engine)
