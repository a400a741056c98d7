"""CPU: host-side logic of the drop-in model class (intake, scaling, constraints, accessors) follows the
reference semantics restated by the oracle (GPSat/models/base_model.py:134-245,
GPSat/models/gpflow_models.py:339-494), plus sharding / synthetic-input helpers."""
import numpy as np
import pandas as pd
import pytest

from gpsat_amd.models import HipGPRModel, get_model
from gpsat_amd import sharding, synthetic as syn
from oracle import gp_oracle as go


class _NoDevice:
    """Stands in for an Engine so that host logic can be constructed without a GPU.  It cannot compute."""
    device_name = "none"
    device_id = 0

    def fit_predict_batch(self, **kw):
        raise RuntimeError("no device in CPU tests")


def _frame(n=40, seed=0):
    rng = np.random.default_rng(seed)
    return pd.DataFrame({"x": rng.uniform(-3e5, 3e5, n), "y": rng.uniform(-3e5, 3e5, n),
                         "t": rng.uniform(-4, 4, n), "z": rng.standard_normal(n) + 2.0})


def test_intake_scaling_matches_reference_semantics():
    df = _frame()
    keep = df.copy()
    m = HipGPRModel(data=df, coords_col=["x", "y", "t"], obs_col="z", coords_scale=[50000, 50000, 1],
                    obs_mean="local", engine=_NoDevice(), expert_loc=np.zeros(3))
    o = go.OracleGPR(df[["x", "y", "t"]].values, df[["z"]].values, coords_scale=[50000, 50000, 1], obs_mean="local")
    np.testing.assert_array_equal(m.coords, o.coords)
    np.testing.assert_array_equal(m.obs, o.obs)
    assert abs(m.obs.mean()) < 1e-12
    pd.testing.assert_frame_equal(df, keep)                      # the caller's frame is not modified
    assert m.param_names == ["lengthscales", "kernel_variance", "likelihood_variance"]
    p = m.get_parameters()
    np.testing.assert_array_equal(p["lengthscales"], np.ones(3))  # gpflow_models.py:129-131
    assert p["kernel_variance"] == 1.0 and p["likelihood_variance"] == 1.0
    # numeric obs_mean is ignored exactly like base_model.py:195-200
    m2 = HipGPRModel(data=df, coords_col=["x", "y", "t"], obs_col="z", obs_mean=5.0, engine=_NoDevice())
    np.testing.assert_array_equal(m2.obs[:, 0], df["z"].values)
    assert m2.coords_col == ["x", "y", "t"] and m2.obs_col == ["z"]


def test_constraints_clamp_and_scale_like_reference():
    df = _frame()
    kw = dict(coords_scale=[50000, 50000, 1])
    m = HipGPRModel(data=df, coords_col=["x", "y", "t"], obs_col="z", engine=_NoDevice(), **kw)
    o = go.OracleGPR(df[["x", "y", "t"]].values, df[["z"]].values, **kw)
    cons = {"lengthscales": {"low": [1e-8, 1e-8, 1e-8], "high": [600000, 600000, 9], "scale": True},
            "likelihood_variance": {"low": 0.5, "high": 1.005}}
    m.set_parameter_constraints(cons, move_within_tol=True, tol=1e-2)     # local_experts.py:1115
    o.set_parameter_constraints(cons, move_within_tol=True, tol=1e-2)
    np.testing.assert_array_equal(m._theta, o.theta)
    np.testing.assert_allclose(m._hi[:3], [12.0, 12.0, 9.0])
    assert m._theta[4] == pytest.approx(1.005 - 1e-2)                      # moved inside [low+tol, high-tol]
    # tol larger than half the narrowest width is shrunk (gpflow_models.py:473-475)
    m.set_kernel_variance_constraints(low=0.9, high=0.91, tol=1.0)
    assert m._theta[3] == pytest.approx(0.905)
    with pytest.raises(AssertionError):
        m.set_lengthscales_constraints(low=[1, 1], high=[2, 2])            # wrong length
    with pytest.raises(AssertionError):
        m.set_parameter_constraints({"nope": {"low": 0, "high": 1}})


def test_accessors_and_errors():
    X = np.random.default_rng(1).uniform(size=(10, 2))
    yv = np.arange(10.0)
    m = HipGPRModel(coords=X, obs=yv, engine=_NoDevice(), kernel="RBF", noise_variance=0.3,
                    kernel_kwargs={"lengthscales": [2.0, 3.0], "variance": 0.7})
    assert m.coords_col == [0, 1] and m.obs_col == [0]
    np.testing.assert_array_equal(m.get_lengthscales(), [2.0, 3.0])
    assert m.get_kernel_variance() == 0.7 and m.get_likelihood_variance() == 0.3
    m.set_parameters(lengthscales=np.array([4.0, 5.0]), kernel_variance=np.array([1.5]), likelihood_variance=0.2)
    assert m.get_parameters("kernel_variance") == {"kernel_variance": 1.5}
    assert m.get_parameters(return_dict=False)[2] == 0.2
    with pytest.warns(UserWarning):
        m.set_likelihood_variance(1e-9)                                    # gpflow_models.py:404-409
    assert m.get_likelihood_variance() == 1e-6
    with pytest.raises(AssertionError):
        m.set_kernel_variance(np.array([1.0, 2.0]))
    with pytest.raises(AssertionError):
        m.set_parameters(bogus=1)
    with pytest.raises(AssertionError):
        HipGPRModel(coords=np.array([[np.nan]]), obs=np.array([1.0]), engine=_NoDevice())
    with pytest.raises(NotImplementedError):
        HipGPRModel(coords=X, obs=yv, engine=_NoDevice(), kernel="Cosine")
    with pytest.raises(NotImplementedError):
        HipGPRModel(coords=np.zeros((4, 5)), obs=np.arange(4.0), engine=_NoDevice())        # built for 1..4 input dimensions
    assert HipGPRModel(coords=np.random.default_rng(0).normal(size=(4, 4)), obs=np.arange(4.0), engine=_NoDevice()).D == 4
    with pytest.raises(NotImplementedError):
        get_model("GPflowSVGPModel")                                        # GPSat/models/__init__.py:24
    assert get_model("GPflowGPRModel") is HipGPRModel
    m._fix_hyperparameters(["kernel_variance", "not_a_param"])
    assert list(m._trainable) == [True, True, False, True]


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(3)
    N = rng.choice([128, 256, 384, 512, 768, 1024, 1536, 2048], size=500)
    P = np.full(500, 500)
    parts = sharding.partition_tiles(N, P, 8)
    allidx = np.sort(np.concatenate(parts))
    np.testing.assert_array_equal(allidx, np.arange(500))
    loads = np.array([sharding.tile_cost(N[p], P[p]).sum() for p in parts])
    assert loads.max() / loads.mean() < 1.05
    for p in parts:
        assert np.all(np.diff(p) > 0)
    # uniform tiles: equal counts
    parts = sharding.partition_tiles(np.full(64, 500), np.full(64, 500), 8)
    assert all(len(p) == 8 for p in parts)
    assert sharding.partition_tiles(N, P, 8)[3].tolist() == sharding.partition_tiles(N, P, 8)[3].tolist()


def test_pack_subset_round_trip():
    b = syn.make_batch(6, [5, 0, 7, 3, 9, 1], [2, 3, 0, 4, 1, 2], D=2, kid=2, base_seed=5)
    idx = np.array([1, 3, 4])
    s = sharding.pack_subset(b, idx)
    assert s["T"] == 3 and s["obs_off"].tolist() == [0, 0, 3, 12] and s["pred_off"].tolist() == [0, 3, 7, 8]
    np.testing.assert_array_equal(s["X"][0:3], b["X"][b["obs_off"][3]:b["obs_off"][4]])
    np.testing.assert_array_equal(s["Xs"][3:7], b["Xs"][b["pred_off"][3]:b["pred_off"][4]])


def test_synthetic_is_deterministic_and_demeaned():
    a = syn.make_batch(3, 50, 10, base_seed=9)
    b = syn.make_batch(3, 50, 10, base_seed=9)
    np.testing.assert_array_equal(a["X"], b["X"])
    np.testing.assert_array_equal(a["y"], b["y"])
    assert abs(a["y"][:50].astype(np.float64).mean()) < 1e-6
    assert np.all(np.linalg.norm(a["Xs"][:, :2], axis=1) <= 4.0 + 1e-5) and np.all(a["Xs"][:, 2] == 0)
