"""CPU: the post-processing oracle (oracle/post_oracle.py) against outputs of the reference's own functions
(tests/golden/ref_post.npz, made by tests/golden/make_golden.py) and against the glued-prediction scores printed in
docs/notebooks/1d_local_expert_model_part_2.ipynb (2 experts: MSE 0.0005, mean log-likelihood 2.5734; 4 experts:
2.7179)."""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.stats

from oracle import gp_oracle as go
from oracle import post_oracle as po

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_post.npz"))


def test_smoothing_oracle_matches_reference_outputs():
    out = po.gaussian_2d_weight(G["sx"], G["sy"], G["sx"], G["sy"], float(G["lx"]), float(G["ly"]), G["svals"])
    np.testing.assert_allclose(out, G["smoothed"], rtol=1e-14, atol=0)
    tiny = po.gaussian_2d_weight(G["sx"], G["sy"], G["sx"], G["sy"], 1.0, 1.0, G["svals"])
    np.testing.assert_array_equal(np.isnan(tiny), np.isnan(G["smoothed_tiny"]))
    np.testing.assert_array_equal(np.isnan(tiny), np.isnan(G["svals"]))          # isolated NaN points stay NaN
    np.testing.assert_allclose(tiny[~np.isnan(tiny)], G["smoothed_tiny"][~np.isnan(tiny)], rtol=1e-14)


def test_glue_oracle_matches_reference_outputs():
    p1 = pd.DataFrame(G["p1"], columns=["x", "pred_loc_x", "f*", "f*_var"])
    g1 = po.glue_local_predictions(p1, "pred_loc_x", "x", ["f*", "f*_var"], float(G["r1"]))
    np.testing.assert_allclose(g1[["pred_loc_x", "f*", "f*_var"]].values, G["g1"], rtol=1e-14)
    p2 = pd.DataFrame(G["p2"], columns=["x", "y", "pred_loc_x", "pred_loc_y", "f*", "f*_var", "y_var"])
    g2 = po.glue_local_predictions(p2, ["pred_loc_x", "pred_loc_y"], ["x", "y"], ["f*", "f*_var", "y_var"], float(G["r2"]))
    np.testing.assert_allclose(g2[["pred_loc_x", "pred_loc_y", "f*", "f*_var", "y_var"]].values, G["g2"], rtol=1e-14)


@pytest.mark.parametrize("radius,locs,mse,mll", [(0.15, [0.25, 0.45], 0.0005, 2.5734),
                                                 (0.10, [0.2, 0.3, 0.4, 0.5], None, 2.7179)])
def test_notebook_glued_scores(radius, locs, mse, mll):
    """oracle GP per expert + oracle gluing reproduce the scores the reference notebook prints (4 decimals)."""
    np.random.seed(0)
    N, noise_std = 100, 0.05
    X_grid = np.linspace(0.1, 0.6, 100)
    X = np.random.uniform(0.1, 0.6, (N,))
    y = np.sin(1 / X) + noise_std * np.random.randn(N)
    f_truth = np.sin(1 / X_grid)
    inference_radius = radius + 1e-8
    rows = []
    for loc in locs:
        m = (X <= loc + radius) & (X >= loc - radius)
        o = go.OracleGPR(X[m, None], y[m, None], kernel="RBF", noise_variance=noise_std ** 2)
        assert o.optimise_parameters(fixed_params=["likelihood_variance"])
        xs = X_grid[np.abs(X_grid - loc) < inference_radius]
        pr = o.predict(xs[:, None])
        rows.append(pd.DataFrame({"x": loc, "pred_loc_x": xs, "f*": pr["f*"], "f*_var": pr["f*_var"]}))
    glued = po.glue_local_predictions(pd.concat(rows), "pred_loc_x", "x", ["f*", "f*_var"], inference_radius)
    assert len(glued) == len(X_grid)
    f_mean, f_std = glued["f*"].values, np.sqrt(glued["f*_var"].values)
    if mse is not None:
        assert f"{np.mean((f_truth - f_mean) ** 2):.4f}" == f"{mse:.4f}"
    assert f"{scipy.stats.norm.logpdf(f_truth, f_mean, f_std).mean():.4f}" == f"{mll:.4f}"
