"""Drop-in protocol test with the REAL reference drivers (`bayes_optim.BO / ParallelBO`), build container only.

`/root/reference` exists only here (never on the GPU box), and here there is no GPU -- so the engine under the
bogp classes is the oracle-backed stand-in of tests/support/oracle_engine.py.  What this proves is the HOST side of
INTEGRATION.md: the reference's ask/tell loop runs unmodified against `bogp.GaussianProcess`, finds the `bogp`
acquisition classes by name, and accepts `bogp.argmax_restart` (optimizer="sweep" and "BFGS") behind its own
`argmax_restart` signature.  Numerical parity of the device path is the job of the `-m gpu` tests."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bayes_optim")), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    for p in (REF, os.path.join(ROOT, "oracle", "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import warnings

    warnings.filterwarnings("ignore")
    import bayes_optim
    import bayes_optim.base as rbase
    import bayes_optim.bayes_opt as ropt

    import bogp
    from support.oracle_engine import OracleEngine

    saved = (rbase.argmax_restart, rbase.AcquisitionFunction, ropt.AcquisitionFunction)
    rbase.argmax_restart = bogp.argmax_restart
    rbase.AcquisitionFunction = ropt.AcquisitionFunction = bogp.acquisition  # section 3
    yield bayes_optim, bogp, OracleEngine
    rbase.argmax_restart, rbase.AcquisitionFunction, ropt.AcquisitionFunction = saved


def _model(bogp, OracleEngine, dim, **kw):
    gp = bogp.GaussianProcess(
        mean=bogp.trend.constant_trend(dim), corr="matern", thetaL=1e-3 * np.ones(dim) * 10, thetaU=1e3 * np.ones(dim) * 10,
        nugget=1e-6, optimizer="BFGS", wait_iter=3, random_start=max(5, dim), eval_budget=100 * dim, **kw)  # fmt: skip
    gp._engine = OracleEngine()
    return gp


@pytest.mark.timeout(600)
@pytest.mark.parametrize("inner", ["sweep", "BFGS"])
def test_reference_BO_runs_on_bogp_classes(ref, inner):
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import BO, RealSpace

    np.random.seed(42)
    dim = 2
    space = RealSpace([-5, 5]) * dim
    model = _model(bogp, OracleEngine, dim)
    aq = {"optimizer": "sweep", "max_FEs": 2000} if inner == "sweep" else {"optimizer": "BFGS", "max_FEs": 100, "n_restart": 3}
    opt = BO(search_space=space, obj_fun=lambda x: float(np.sum(np.asarray(x) ** 2)), model=model, DoE_size=5,
             max_FEs=14, verbose=False, n_point=1, acquisition_fun="EI", acquisition_optimization=aq, random_seed=42)  # fmt: skip
    xopt, fopt, _ = opt.run()
    assert opt.eval_count == 14 and model.is_fitted and len(xopt) == dim
    assert np.isfinite(fopt) and fopt <= 50.0  # plumbing test: optimisation quality is covered on the GPU (examples/)
    assert type(model).__module__.startswith("bogp")


@pytest.mark.timeout(600)
def test_reference_ParallelBO_with_mgfi(ref):
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import ParallelBO, RealSpace

    np.random.seed(1)
    dim = 3
    space = RealSpace([-5, 5]) * dim
    model = _model(bogp, OracleEngine, dim)
    opt = ParallelBO(search_space=space, obj_fun=lambda x: float(np.sum(np.asarray(x) ** 2)), model=model, DoE_size=6,
                     max_FEs=15, verbose=False, n_point=3, acquisition_fun="MGFI", acquisition_par={"t": 2},
                     acquisition_optimization={"optimizer": "sweep", "max_FEs": 1500}, random_seed=1)  # fmt: skip
    X = opt.ask()
    assert len(X) == 6  # DoE
    opt.tell(X, [float(np.sum(np.asarray(x) ** 2)) for x in X])
    X = opt.ask()
    assert len(X) == 3 and model.is_fitted
    opt.tell(X, [float(np.sum(np.asarray(x) ** 2)) for x in X])
    assert opt.eval_count == 9


class _CountingEngine:
    """OracleEngine that counts posterior passes (predict / sweep / sweep_topk each walk all candidates once)."""

    def __new__(cls):
        from support.oracle_engine import OracleEngine

        class Counting(OracleEngine):
            passes = 0

            def predict(self, eval_MSE=True):
                type(self).passes += 1
                return super().predict(eval_MSE)

        return Counting()


@pytest.mark.timeout(600)
def test_install_fuses_the_parallelbo_batch_into_one_posterior_pass(ref):
    """SURVEY row f1 / VERDICT r01 item 4: through the REAL `ParallelBO.ask()` with n_point = 8, `bogp.install` makes the
    q criteria share ONE posterior pass (the reference's loop costs 8), the proposals are 8 distinct points none of which
    repeats an evaluated one.  (The t_i are drawn by the reference's own sampler before any candidate is sampled, in both
    paths: bayes_opt.py:101-106.)"""
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import ParallelBO, RealSpace

    dim, q = 3, 8
    f = lambda x: float(np.sum(np.asarray(x) ** 2))  # noqa: E731

    def run(fused):
        undo = bogp.install(bayes_optim, fuse_batch=fused)
        try:
            np.random.seed(11)
            model = _model(bogp, OracleEngine, dim)
            model._engine = _CountingEngine()
            opt = ParallelBO(search_space=RealSpace([-5, 5]) * dim, obj_fun=f, model=model, DoE_size=10, max_FEs=40, verbose=False,
                             n_point=q, acquisition_fun="MGFI", acquisition_par={"t": 2},
                             acquisition_optimization={"optimizer": "sweep", "max_FEs": 3000}, random_seed=11)  # fmt: skip
            X0 = opt.ask()
            opt.tell(X0, [f(x) for x in X0])
            np.random.seed(5)
            before = type(model._engine).passes
            X = opt.ask()
            passes = type(model._engine).passes - before
            return X, passes, np.asarray(opt.data, dtype=float)[:, :dim]
        finally:
            undo()

    Xf, passes_f, hist = run(True)
    Xr, passes_r, _ = run(False)
    assert passes_f == 1 and passes_r == q
    Xf = np.asarray(Xf, dtype=float)
    assert Xf.shape == (q, dim)
    assert len({tuple(np.round(x, 12)) for x in Xf}) == q  # distinct proposals: fall-backs instead of random padding
    assert not any(np.any(np.all(np.isclose(hist, x), axis=1)) for x in Xf)


def test_install_reroutes_the_default_bfgs_to_one_sweep(ref):
    """A driver built WITHOUT `acquisition_optimization` gets the reference's default inner optimiser "BFGS" with 100 dim point
    evaluations (base.py:200-214): `install(reroute_bfgs="sweep", sweep_budget=...)` turns its ask() into one sweep -- same
    constructor call, one posterior pass for the 4 proposals; `uninstall()` clears the reroute."""
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import ParallelBO, RealSpace

    from bogp import integration

    dim, q = 2, 4
    f = lambda x: float(np.sum(np.asarray(x) ** 2))  # noqa: E731
    undo = bogp.install(bayes_optim, reroute_bfgs="sweep", sweep_budget=2500)
    try:
        np.random.seed(3)
        model = _model(bogp, OracleEngine, dim)
        model._engine = _CountingEngine()
        opt = ParallelBO(search_space=RealSpace([-5, 5]) * dim, obj_fun=f, model=model, DoE_size=8, max_FEs=30, verbose=False,
                         n_point=q, acquisition_fun="MGFI", acquisition_par={"t": 2}, random_seed=3)  # fmt: skip
        assert opt._optimizer == "BFGS"  # the reference's own default: nothing was passed
        X0 = opt.ask()
        opt.tell(X0, [f(x) for x in X0])
        before = type(model._engine).passes
        X = opt.ask()
        assert type(model._engine).passes - before == 1
        assert np.asarray(X, dtype=float).shape == (q, dim)
        assert model._engine.M == 2500  # the sweep's own budget, not BFGS's 100 dim
    finally:
        undo()
    assert not integration._REROUTE
    with pytest.raises(ValueError):
        bogp.install(bayes_optim, reroute_bfgs="CMA")
    bogp.uninstall()


@pytest.mark.timeout(600)
def test_fused_batch_with_fixed_variables_and_ucb(ref):
    """ask(fixed=...) through the fused batch (the free variables are swept, the fixed one is filled in for the model and
    by the reference's `fillin_fixed_value` afterwards) and the UCB sampler (alpha_i logit-normal, bayes_opt.py:87-90)."""
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import ParallelBO, RealSpace

    dim = 3
    f = lambda x: float(np.sum(np.asarray(x) ** 2))  # noqa: E731
    undo = bogp.install(bayes_optim)
    try:
        np.random.seed(2)
        model = _model(bogp, OracleEngine, dim)
        model._engine = _CountingEngine()
        opt = ParallelBO(search_space=RealSpace([-5, 5]) * dim, obj_fun=f, model=model, DoE_size=8, max_FEs=30, verbose=False,
                         n_point=4, acquisition_fun="UCB", acquisition_par={"alpha": 0.5},
                         acquisition_optimization={"optimizer": "sweep", "max_FEs": 1000}, random_seed=2)  # fmt: skip
        X0 = opt.ask()
        opt.tell(X0, [f(x) for x in X0])
        before = type(model._engine).passes
        name = opt.search_space.var_name[1]
        X = np.asarray(opt.ask(fixed={name: 1.25}), dtype=float)
        assert type(model._engine).passes - before == 1
        assert X.shape == (4, dim) and np.all(X[:, 1] == 1.25)
        assert len({tuple(x) for x in X}) == 4
    finally:
        undo()


@pytest.mark.timeout(300)
def test_save_load_roundtrip_through_dill(ref, tmp_path):
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import BO, RealSpace

    np.random.seed(3)
    dim = 2
    model = _model(bogp, OracleEngine, dim)
    opt = BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(np.sum(np.asarray(x) ** 2)), model=model,
             DoE_size=5, max_FEs=7, verbose=False, n_point=1, acquisition_fun="EI",
             acquisition_optimization={"optimizer": "sweep", "max_FEs": 500}, random_seed=3)  # fmt: skip
    opt.step()
    f = str(tmp_path / "opt.pkl")
    opt.save(f)  # base.py:499-519 dills the whole optimiser, model included
    opt2 = BO.load(f)
    assert opt2.model._engine is None  # device/engine handles never travel
    assert opt2.model.is_fitted and np.allclose(opt2.model.theta_, model.theta_)


@pytest.mark.timeout(600)
def test_reference_BO_with_a_linear_trend_model(ref):
    """Universal kriging (linear_trend, p = d + 1): the host layer's matrix-shaped trend state (Ft, G, Q, beta) under the
    reference's own BO loop with the gradient-based inner optimiser (which calls model.gradient through EI.return_dx)."""
    bayes_optim, bogp, OracleEngine = ref
    from bayes_optim import BO, RealSpace

    np.random.seed(7)
    dim = 2
    gp = bogp.GaussianProcess(mean=bogp.trend.linear_trend(dim), corr="squared_exponential", thetaL=[1e-2] * dim, thetaU=[1e2] * dim,
                              nugget=1e-6, random_start=3, eval_budget=60)  # fmt: skip
    gp._engine = OracleEngine()
    opt = BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(np.sum(np.asarray(x) ** 2) + x[0]), model=gp, DoE_size=6,
             max_FEs=10, verbose=False, n_point=1, acquisition_fun="EI",
             acquisition_optimization={"optimizer": "BFGS", "max_FEs": 60, "n_restart": 2}, random_seed=7)  # fmt: skip
    opt.run()
    assert opt.eval_count == 10 and gp.is_fitted
    assert gp.Ft.shape == (gp.X.shape[0], dim + 1) and gp.G.shape == (dim + 1, dim + 1) and np.ravel(gp.mean.beta).shape == (dim + 1,)
    mu, mse = gp.predict(np.zeros((3, dim)), eval_MSE=True)
    assert mu.shape == (3, 1) and np.all(mse >= 0)
    dmu, dmse = gp.gradient(np.zeros((1, dim)))
    assert dmu.shape == (dim, 1) and dmse.shape == (dim, 1)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kw", [dict(nugget=1e-6), dict(nugget=1e-6, noise_estim=True)])
def test_fit_with_the_restricted_likelihood_host_logic(kw):
    """REML through the host layer (parameter lists of gpr.py:1073-1086, the NOISY-mode commit, env, attribute refresh),
    with the oracle as the engine.  The reference's own fit(likelihood="restricted") raises TypeError at gpr.py:405."""
    import bogp
    from oracle import gp_oracle as O
    from support.oracle_engine import OracleEngine

    rng = np.random.default_rng(3)
    X = rng.uniform(-5, 5, size=(30, 2))
    y = (np.sum(X**2, axis=1) + rng.standard_normal(30)).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(2), corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2,
                              likelihood="restricted", random_start=2, eval_budget=80, **kw)  # fmt: skip
    gp._engine = OracleEngine()
    np.random.seed(5)
    assert gp.fit(X, y) is gp and gp.is_fitted
    names = ["theta", "sigma2"] + (["noise_var"] if kw.get("noise_estim") else [])
    assert list(gp.par) == names
    par = np.concatenate([np.ravel(gp.par[k]) for k in names])
    mode = O.MODE_NOISE_ESTIM if kw.get("noise_estim") else O.MODE_NOISY
    ref = O.log_likelihood_restricted(par, X, y, O.KERNEL_MATERN32, mode, noise_var=1e-6, estimate_trend=True, beta=None)
    assert gp.log_likelihood_ == ref and np.isfinite(ref)
    assert gp.sigma2.shape == (1,) and float(gp.sigma2[0]) == float(gp.par["sigma2"][0])
    mu, mse = gp.predict(X[:4], eval_MSE=True)
    assert mu.shape == (4, 1) and np.all(mse >= 0)


@pytest.mark.timeout(600)
def test_multitarget_fit_host_logic():
    """Y (N, 3) through the host layer (shapes of sigma2 / gamma / rho / Yt, the (M, 3) posterior) with the oracle as the
    engine; the device path is checked against the reference's own outputs in tests/test_gpu_parity.py (G17)."""
    import bogp
    from oracle import gp_oracle as O
    from support.oracle_engine import OracleEngine

    rng = np.random.default_rng(4)
    X = rng.uniform(-5, 5, size=(30, 2))
    Y = np.c_[np.sum(X**2, axis=1), np.sin(X[:, 0]) + X[:, 1]] + 0.3 * rng.standard_normal((30, 2))
    Y = (Y - Y.mean(0)) / Y.std(0)
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(2, beta=0.0), corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2,
                              nugget=1e-3, random_start=2, eval_budget=80)  # fmt: skip
    gp._engine = OracleEngine()
    np.random.seed(5)
    assert gp.fit(X, Y) is gp and gp.is_fitted
    par = np.r_[gp.theta_, gp.sigma2[0]]
    assert gp.log_likelihood_ == O.log_likelihood_concentrated(par, X, Y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-3, beta=0.0)
    assert gp.sigma2.shape == (2,) and gp.gamma.shape == (30, 2) and gp.rho.shape == (30, 2) and gp.Yt.shape == (30, 2)
    mu, mse = gp.predict(X[:4], eval_MSE=True)
    st = O.make_state(par, X, Y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-3, beta=0.0)
    omu, omse = O.predict(st, X[:4])
    np.testing.assert_array_equal(mu, omu)
    np.testing.assert_array_equal(mse, omse)
    with pytest.raises(NotImplementedError):
        bogp.GaussianProcess(mean=bogp.trend.constant_trend(2), corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, nugget=1e-3).fit(X, Y)
