"""CPU ORACLE (test infrastructure only) for the post-processing rows of SURVEY.md section 8f:
hyper-parameter smoothing (GPSat/postprocessing.py:22-52 ``gaussian_2d_weight``, numba guvectorize) and gluing of
overlapping local predictions (GPSat/postprocessing.py:447-577).  NumPy / pandas restatements; pinned in
tests/test_post_cpu.py by the glued-prediction scores printed in docs/notebooks/1d_local_expert_model_part_2.ipynb
(MSE 0.0005 / mean log-likelihood 2.5734 for 2 experts, 2.7179 for 4 experts)."""
import numpy as np
import pandas as pd
from scipy.stats import norm


def gaussian_2d_weight(x0, y0, x, y, l_x, l_y, vals):
    """For every reference position (x0[i], y0[i]): sum_j w_ij vals_j / sum_j w_ij over the non-NaN vals,
    w_ij = exp(-d2/2), d2 = ((x_j-x0_i)/l_x)^2 + ((y_j-y0_i)/l_y)^2; NaN when the weights sum to zero
    (postprocessing.py:32-52; accumulation in index order like the reference loop)."""
    x0, y0, x, y, vals = (np.asarray(v, dtype=np.float64) for v in (x0, y0, x, y, vals))
    out = np.empty(len(x0))
    ok = ~np.isnan(vals)
    for i in range(len(x0)):
        d2 = ((x - x0[i]) / l_x) ** 2 + ((y - y0[i]) / l_y) ** 2
        w = np.exp(-d2 / 2)
        w_val = 0.0
        w_sum = 0.0
        for j in np.nonzero(ok)[0]:
            w_val += w[j] * vals[j]
            w_sum += w[j]
        out[i] = np.nan if w_sum == 0 else w_val / w_sum
    return out


def glue_local_predictions(preds_df, pred_loc_cols, xprt_loc_cols, vars_to_glue, inference_radius, R=3):
    """postprocessing.py:447-577 (1-D and 2-D variants are the same formula with one or two coordinate columns)."""
    if isinstance(pred_loc_cols, str):
        pred_loc_cols, xprt_loc_cols = [pred_loc_cols], [xprt_loc_cols]
    if isinstance(vars_to_glue, str):
        vars_to_glue = [vars_to_glue]
    preds = preds_df.copy(deep=True)
    preds["total_weights"] = 1.0
    if isinstance(inference_radius, dict):
        inference_radius = np.array([inference_radius[loc] for loc in preds[xprt_loc_cols[0]]])
    for pc, xc in zip(pred_loc_cols, xprt_loc_cols):
        preds["total_weights"] *= norm.pdf(preds[pc], preds[xc], inference_radius / R)
    for var in vars_to_glue:
        preds[var] = preds[var] * preds["total_weights"]
    glued = preds[list(pred_loc_cols) + ["total_weights"] + vars_to_glue].groupby(list(pred_loc_cols)).sum().reset_index()
    for var in vars_to_glue:
        glued[var] = glued[var] / glued["total_weights"]
    return glued.drop("total_weights", axis=1)
