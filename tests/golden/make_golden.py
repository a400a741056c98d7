"""
Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Sources of truth (none of them travels to the GPU box; only the .npz outputs do):

1. ``kat_sklearn_matern32.npz`` -- the reference's own known-answer fixture,
   tests/test_localexperts.py:22-49 replayed verbatim with scikit-learn
   (the oracle the reference's test_gpflow_gpr compares GPflowGPRModel against).
2. ``ref_purepython_matern32_N*.npz`` -- outputs of the REFERENCE's NumPy functions
   SGPkernel / SMLII_mod / GPR (GPSat/models/pure_python_gpr.py:378-553), imported
   from /root/reference.  Unrelated heavy imports of the package (tensorflow, tables,
   numba, pyproj, deprecated) are absent in this container and are satisfied by inert
   placeholder modules; only the three pure NumPy/SciPy functions are executed.
3. ``kat_notebook_rbf.npz`` -- inputs regenerated from the seeds printed in
   docs/notebooks/gp_regression.ipynb (cell 3) together with the values that notebook
   prints (LML 16.6180 -> 21.4700, lengthscale 1.5648, kernel_variance 0.5168).
4. ``transforms.npz`` -- softplus / sigmoid tables on the grids of tests/test_utils.py:962-1023: extended-precision
   closed forms AND the outputs of the reference's own ``softplus`` / ``sigmoid`` / ``_inverse_softplus`` /
   ``_inverse_sigmoid`` (GPSat/utils.py:2320-2400; numba is absent, the inert decorator leaves the plain Python bodies
   of the two inverse kernels, called element by element).
6. ``kat_sklearn_kernels.npz`` -- scikit-learn (the oracle of the reference's own test, tests/test_localexperts.py:40-49)
   for the covariance functions that test does not cover: Matern-1/2, Matern-5/2 and RBF with ARD length scales in
   D = 3, N in {50, 500}: log marginal likelihood and its gradient at two parameter vectors, predictive mean / std /
   full covariance, and (N = 50) the optimum sklearn's own L-BFGS-B finds.
5. ``ref_post.npz`` -- outputs of the REFERENCE's gaussian_2d_weight and glue_local_predictions_1d/_2d
   (GPSat/postprocessing.py) on seeded inputs (``python tests/golden/make_golden.py post`` regenerates only this one).

Fixtures are DATA (inputs + expected outputs), never reference source text.
"""
import os
import sys
import types
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def kat_sklearn():
    from sklearn.gaussian_process.kernels import Matern
    from sklearn.gaussian_process import GaussianProcessRegressor
    np.random.seed(23435)
    kernel = Matern(length_scale=0.8, nu=3 / 2)
    gp = GaussianProcessRegressor(kernel)
    x = np.linspace(0, 10, 100)[:, None]
    f = gp.sample_y(x, random_state=0)
    N = 50
    eps = 1e-2
    indices = np.arange(100)
    np.random.shuffle(indices)
    x_train = x[indices[:N]]
    y_train = f[indices[:N]] + eps * np.random.randn(N, 1)
    gp.alpha = eps ** 2
    gp.fit(x_train, y_train)
    ls = gp.kernel_.length_scale
    ml = gp.log_marginal_likelihood()
    test_index = np.random.randint(0, 99)
    x_test = x[[test_index]]
    pred_mean, pred_std = gp.predict(x_test, return_std=True)
    # full posterior covariance at a few grid points (sklearn return_cov: the reference's sklearnGPRModel full_cov path)
    cov_index = np.array([5, 40, 41, 42, 77, 93])
    cov_mean, cov = gp.predict(x[cov_index], return_cov=True)
    np.savez(os.path.join(HERE, "kat_sklearn_matern32.npz"), cov_index=cov_index, cov_x=x[cov_index, 0], cov=cov,
             cov_mean=np.ravel(cov_mean),
             x_train=x_train[:, 0], y_train=y_train[:, 0], eps=eps, ls=ls, ml=ml,
             test_index=test_index, x_test=x_test[0, 0],
             pred_mean=np.ravel(pred_mean)[0], pred_std=np.ravel(pred_std)[0])
    print("sklearn KAT: ls", ls, "ml", ml, "x*", x_test[0, 0], "mean", pred_mean, "std", pred_std)


class _Inert(types.ModuleType):
    """Placeholder for absent third-party modules: attribute access returns an inert
    object that is callable / decorator-transparent / subscriptable / iterable-as-empty."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj()


class _InertObj:
    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _InertObj()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj()

    def __getitem__(self, k):
        return _InertObj()

    def __iter__(self):
        return iter(())


def ref_purepython():
    for name in ["tensorflow", "tensorflow.python", "tensorflow.python.client", "tables", "numba",
                 "pyproj", "deprecated"]:
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from GPSat.models.pure_python_gpr import SGPkernel, SMLII_mod, GPR
    for N in (16, 128, 500):
        rng = np.random.default_rng(1000 + N)
        D = 3
        x = np.column_stack([rng.uniform(-6, 6, N), rng.uniform(-6, 6, N), rng.uniform(-4, 4, N)])
        ell = np.array([3.0, 4.5, 5.0])
        sf2, sn2 = 0.7, 0.05
        K = SGPkernel(x, ell=ell, sigma=sf2) + sn2 * np.eye(N)
        y = np.linalg.cholesky(K) @ rng.standard_normal(N)
        P = 24
        xs = np.column_stack([rng.uniform(-4, 4, P), rng.uniform(-4, 4, P), np.zeros(P)])
        thetas = np.array([[3.0, 4.5, 5.0, 0.7, 0.05],
                           [1.0, 1.0, 1.0, 1.0, 1.0],
                           [2.0, 7.0, 3.5, 0.3, 0.01]])
        nll = np.array([float(np.squeeze(SMLII_mod(hypers=th, x=x, y=y, grad=False))) for th in thetas])
        # gradient goldens: central differences of the reference NLL (fp64), w.r.t. theta
        grads = np.zeros_like(thetas)
        for i, th in enumerate(thetas):
            for j in range(5):
                h = 1e-5 * max(1.0, abs(th[j]))
                tp, tm = th.copy(), th.copy()
                tp[j] += h
                tm[j] -= h
                grads[i, j] = (float(np.squeeze(SMLII_mod(hypers=tp, x=x, y=y, grad=False))) -
                               float(np.squeeze(SMLII_mod(hypers=tm, x=x, y=y, grad=False)))) / (2 * h)
        means, stds = [], []
        for th in thetas:
            m, s = GPR(x, y[:, None], xs, th[:3], th[3], th[4], mean=0)
            means.append(np.ravel(m))
            stds.append(np.ravel(s))
        Kxx = SGPkernel(x[:8], ell=ell, sigma=sf2)
        np.savez(os.path.join(HERE, f"ref_purepython_matern32_N{N}.npz"),
                 x=x, y=y, xs=xs, thetas=thetas, nll=nll, grads_fd=grads,
                 pred_mean=np.array(means), pred_std=np.array(stds), K8=Kxx)
        print(f"ref pure-python N={N}: nll {nll}")


def kat_notebook_rbf():
    np.random.seed(0)
    N, L, noise_std = 30, 5, 0.05
    X_grid = np.linspace(-L, L, 100)
    X = np.random.uniform(-L, L, (N,))
    epsilon = noise_std * np.random.randn(N)
    y = np.cos(X) + epsilon
    np.savez(os.path.join(HERE, "kat_notebook_rbf.npz"), X=X, y=y, X_grid=X_grid,
             f_truth=np.cos(X_grid),
             # printed by docs/notebooks/gp_regression.ipynb (sklearnGPRModel: amplitude = sqrt(printed kv),
             # GPSat/models/sklearn_models.py:96)
             lml_init=16.6180, kv_printed_init=1.5, lik_var=0.0025, ls_init=1.0,
             mse_init=0.0026, mll_init=1.4578,
             ls_opt=1.5648, kv_printed_opt=0.5168, lml_opt=21.4700, mse_opt=0.0037, mll_opt=1.8717)
    print("notebook KAT data written")


def _inert_imports():
    """absent third-party roots resolve to inert placeholder packages; GPSat itself is imported from /root/reference"""
    import importlib.abc
    import importlib.machinery
    roots = {"tensorflow", "tables", "numba", "pyproj", "deprecated", "xarray", "netCDF4", "dataclasses_json", "gpflow",
             "astropy", "global_land_mask", "matplotlib", "cartopy", "seaborn", "gpytorch", "chardet", "shapely"}

    class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, fullname, path, target=None):
            if fullname.split(".")[0] in roots:
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            return _Inert(spec.name)

        def exec_module(self, module):
            module.__path__ = []

    if not any(type(f).__name__ == "_InertFinder" for f in sys.meta_path):
        for k in [k for k in sys.modules if k.split(".")[0] in roots]:
            del sys.modules[k]
        sys.meta_path.insert(0, _InertFinder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def transforms():
    x1 = np.linspace(-100, 100, 1000)
    x2 = np.linspace(-10, 10, 1000)
    # closed forms in extended precision as the independent check of the fp64 restatement
    xl = x1.astype(np.longdouble)
    sp = np.where(xl > 0, xl + np.log1p(np.exp(-xl)), np.log1p(np.exp(xl))).astype(np.float64)
    x2l = x2.astype(np.longdouble)
    sg = (1 / (1 + np.exp(-x2l))).astype(np.float64)
    # the reference's own functions (GPSat/utils.py:2320-2400)
    _inert_imports()
    from GPSat import utils as U
    ref_sp = U.softplus(x1)
    ref_sp_shift = U.softplus(x1, shift=10.0)
    ref_sg = U.sigmoid(x2)
    ref_sg_box = U.sigmoid(x2, -1.0, 2.5)

    def elementwise(body, ys, *consts):
        out = np.empty(len(ys))
        o = np.empty(1)
        for i, v in enumerate(ys):
            body(np.array([v]), *[np.array([c], dtype=np.float64) for c in consts], o)
            out[i] = o[0]
        return out
    thr = np.log(np.finfo(np.float64).eps) + 2.0
    y_sp = np.concatenate([ref_sp, [-1.0, 0.0, 1e-300, 1e-20, 40.0, 800.0]])
    ref_isp = elementwise(U._inverse_softplus, y_sp, 0.0, thr)
    y_sps = np.concatenate([ref_sp_shift, [9.0, 10.0, 10.0 + 1e-9]])
    ref_isp_shift = elementwise(U._inverse_softplus, y_sps, 10.0, thr)
    y_sg = np.concatenate([ref_sg_box, [-1.5, -1.0, 2.5, 3.0]])
    ref_isg_box = elementwise(U._inverse_sigmoid, y_sg, -1.0, 2.5)
    np.savez(os.path.join(HERE, "transforms.npz"), x_softplus=x1, softplus=sp, x_sigmoid=x2, sigmoid=sg,
             ref_softplus=ref_sp, ref_softplus_shift10=ref_sp_shift, ref_sigmoid=ref_sg, ref_sigmoid_box=ref_sg_box,
             box=np.array([-1.0, 2.5]), y_inv_softplus=y_sp, ref_inv_softplus=ref_isp, y_inv_softplus_shift10=y_sps,
             ref_inv_softplus_shift10=ref_isp_shift, y_inv_sigmoid_box=y_sg, ref_inv_sigmoid_box=ref_isg_box)
    print("transform tables written (closed forms + reference functions)")


def kat_sklearn_kernels():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern
    out = {}
    kinds = {"Matern12": lambda ls: Matern(length_scale=ls, nu=0.5), "Matern52": lambda ls: Matern(length_scale=ls, nu=2.5),
             "RBF": lambda ls: RBF(length_scale=ls)}
    for N in (50, 500):
        rng = np.random.default_rng(4200 + N)
        X = np.column_stack([rng.uniform(-6, 6, N), rng.uniform(-6, 6, N), rng.uniform(-4, 4, N)])
        Xs = np.column_stack([rng.uniform(-4, 4, 12), rng.uniform(-4, 4, 12), rng.uniform(-1, 1, 12)])
        f = np.sin(0.6 * X[:, 0]) * np.cos(0.4 * X[:, 1]) + 0.3 * np.sin(0.5 * X[:, 2])
        y = f + 0.1 * rng.standard_normal(N)
        y = y - y.mean()
        out[f"X_{N}"], out[f"y_{N}"], out[f"Xs_{N}"] = X, y, Xs
        thetas = np.array([[2.5, 4.0, 3.0, 0.6, 0.02], [1.0, 1.0, 1.0, 1.0, 1.0]])      # (l1, l2, l3, sf2, sn2)
        out[f"thetas_{N}"] = thetas
        for name, mk in kinds.items():
            lml, grads, means, stds, covs = [], [], [], [], []
            for th in thetas:
                k = ConstantKernel(th[3]) * mk(th[:3].copy())
                gp = GaussianProcessRegressor(kernel=k, alpha=th[4], optimizer=None).fit(X, y)
                v, g = gp.log_marginal_likelihood(gp.kernel_.theta, eval_gradient=True)   # gradient w.r.t. log(sf2, l1, l2, l3)
                lml.append(v)
                grads.append(g)
                m, c = gp.predict(Xs, return_cov=True)
                _, sd = gp.predict(Xs, return_std=True)
                means.append(m); stds.append(sd); covs.append(c)
            out[f"{name}_{N}_lml"] = np.array(lml)
            out[f"{name}_{N}_dlml_dlog"] = np.array(grads)
            out[f"{name}_{N}_mean"] = np.array(means)
            out[f"{name}_{N}_std"] = np.array(stds)
            out[f"{name}_{N}_cov"] = np.array(covs)
            if N == 50:
                # sklearn's own optimum (L-BFGS-B on the log parameters, no restarts), noise fixed through alpha
                k = ConstantKernel(1.0, (1e-3, 1e3)) * mk(np.ones(3))
                k.k2.length_scale_bounds = (1e-2, 1e2)
                gp = GaussianProcessRegressor(kernel=k, alpha=0.01, n_restarts_optimizer=0).fit(X, y)
                out[f"{name}_50_opt_theta"] = np.concatenate([gp.kernel_.k2.length_scale, [gp.kernel_.k1.constant_value, 0.01]])
                out[f"{name}_50_opt_lml"] = gp.log_marginal_likelihood_value_
                m, sd = gp.predict(Xs, return_std=True)
                out[f"{name}_50_opt_mean"], out[f"{name}_50_opt_std"] = m, sd
            print(f"sklearn {name} N={N}: lml {lml}")
    np.savez(os.path.join(HERE, "kat_sklearn_kernels.npz"), **out)




def ref_postprocessing():
    """``ref_post.npz`` -- outputs of the REFERENCE's gaussian_2d_weight body (GPSat/postprocessing.py:28-52; numba
    is absent, the inert decorator leaves the plain Python function, called once per reference position) and of
    glue_local_predictions_1d / _2d (:447-577) on seeded inputs."""
    import pandas as pd
    _inert_imports()
    from GPSat.postprocessing import gaussian_2d_weight, glue_local_predictions_1d, glue_local_predictions_2d
    rng = np.random.default_rng(77)
    T = 150
    x = rng.uniform(-1e6, 1e6, T)
    y = rng.uniform(-1e6, 1e6, T)
    vals = np.exp(rng.normal(0, 1, T))
    vals[rng.choice(T, 17, replace=False)] = np.nan
    lx, ly = 2.0e5, 3.0e5

    def smooth(x, y, vals, lx, ly):
        out = np.empty(len(x))
        o = np.empty(1)
        for i in range(len(x)):
            gaussian_2d_weight(x[i], y[i], x, y, np.array([lx]), np.array([ly]), vals, o)
            out[i] = o[0]
        return out
    sm = smooth(x, y, vals, lx, ly)
    # tiny length scales: isolated NaN points get zero total weight -> NaN out
    sm_tiny = smooth(x, y, vals, 1.0, 1.0)
    # gluing, 1-D: 6 experts on a line, prediction grid within the inference radius
    r1 = 0.15 + 1e-8
    rows = []
    grid = np.linspace(0.1, 0.9, 161)
    for e in np.linspace(0.2, 0.8, 6):
        g = grid[np.abs(grid - e) < r1]
        rows.append(pd.DataFrame({"x": e, "pred_loc_x": g, "f*": np.sin(1 / g) + 0.05 * rng.standard_normal(len(g)),
                                  "f*_var": rng.uniform(1e-4, 1e-2, len(g))}))
    p1 = pd.concat(rows).sample(frac=1.0, random_state=3).reset_index(drop=True)
    g1 = glue_local_predictions_1d(p1, "pred_loc_x", "x", ["f*", "f*_var"], r1)
    # 2-D
    r2 = 300.0
    rows = []
    gx, gy = np.meshgrid(np.arange(-500, 501, 50.0), np.arange(-500, 501, 50.0))
    gx, gy = gx.ravel(), gy.ravel()
    for ex in (-250.0, 0.0, 250.0):
        for ey in (-250.0, 0.0, 250.0):
            m = (gx - ex) ** 2 + (gy - ey) ** 2 < r2 ** 2
            n = int(m.sum())
            rows.append(pd.DataFrame({"x": ex, "y": ey, "pred_loc_x": gx[m], "pred_loc_y": gy[m],
                                      "f*": rng.standard_normal(n), "f*_var": rng.uniform(0.01, 1, n),
                                      "y_var": rng.uniform(1, 2, n)}))
    p2 = pd.concat(rows).sample(frac=1.0, random_state=4).reset_index(drop=True)
    g2 = glue_local_predictions_2d(p2, ["pred_loc_x", "pred_loc_y"], ["x", "y"], ["f*", "f*_var", "y_var"], r2)
    np.savez(os.path.join(HERE, "ref_post.npz"), sx=x, sy=y, svals=vals, lx=lx, ly=ly, smoothed=sm, smoothed_tiny=sm_tiny,
             r1=r1, p1=p1[["x", "pred_loc_x", "f*", "f*_var"]].values, g1=g1[["pred_loc_x", "f*", "f*_var"]].values,
             r2=r2, p2=p2[["x", "y", "pred_loc_x", "pred_loc_y", "f*", "f*_var", "y_var"]].values,
             g2=g2[["pred_loc_x", "pred_loc_y", "f*", "f*_var", "y_var"]].values)
    print("ref post-processing: smoothed NaNs", int(np.isnan(sm).sum()), int(np.isnan(sm_tiny).sum()),
          "glued rows", len(g1), len(g2))


if __name__ == "__main__":
    only = sys.argv[1:]
    steps = {"sklearn": kat_sklearn, "notebook": kat_notebook_rbf, "purepython": ref_purepython,
             "transforms": transforms, "kernels": kat_sklearn_kernels, "post": ref_postprocessing}
    for name, fn in steps.items():          # purepython registers plain stubs; the finder-based imports come after it
        if not only or name in only:
            fn()
