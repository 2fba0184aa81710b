"""`bogp.install()` must not take anything away from the reference: the scenarios of the
reference's own `unittest/test_fmin.py:9-28`, `unittest/test_constraint.py:31-90` and a CMA-inner-optimiser driver run
UNDER install(), next to drivers whose model is the reference's CPU GaussianProcess / RandomForest.

Build container only (`/root/reference` is absent on the GPU box); no GPU here, so the engine under `bogp.GaussianProcess`
is the oracle-backed stand-in of tests/support/oracle_engine.py, injected by monkeypatching `bogp._lib.Engine`.  The device
twin is `tests/test_gpu_driver.py::test_fmin_engine_trace_replays_on_the_device` (a recorded `fmin` engine trace)."""
import os
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "bayes_optim")), reason="reference tree not present")


@pytest.fixture()
def installed(monkeypatch):
    for p in (REF, os.path.join(ROOT, "oracle", "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    warnings.filterwarnings("ignore")
    import bayes_optim

    import bogp
    from support.oracle_engine import OracleEngine

    created = []

    class Recording(OracleEngine):
        """remembers the size of every candidate upload (a sweep shows up as one upload of its budget)"""

        def upload_candidates(self, Xs, lazy=False):
            self.__dict__.setdefault("uploads", []).append(len(Xs))
            return super().upload_candidates(Xs)

    def engine(device=0):
        created.append(Recording(device))
        return created[-1]

    monkeypatch.setattr(bogp._lib, "Engine", engine)
    undo = bogp.install(bayes_optim)
    yield bayes_optim, bogp, created
    undo()


def _criterion_behind(w):
    """the acquisition object inside `partial_argument(functools.partial(criterion, return_dx=...))` (base.py:489-494)"""
    import functools

    for _ in range(8):
        if isinstance(w, functools.partial):
            w = w.func
        elif hasattr(w, "__wrapped__"):
            w = w.__wrapped__
        else:
            break
    return w


def _sphere(x):
    x = np.asarray(x)
    return np.sum(x**2)


@pytest.mark.timeout(900)
def test_fmin_runs_on_the_device_classes_after_install(installed):
    """unittest/test_fmin.py:9-28 verbatim in effect; the model `fmin` builds is bogp's (so is its EI), the default inner
    optimiser stays the reference's "BFGS" loop, now delegated to the reference's own function."""
    bayes_optim, bogp, created = installed
    seen = []
    orig_fit = bogp.GaussianProcess.fit

    def spy(self, X, y):
        seen.append(type(self))
        return orig_fit(self, X, y)

    bogp.GaussianProcess.fit = spy
    try:
        minimum = bayes_optim.fmin(_sphere, [-5] * 2, [5] * 2, seed=42, max_FEs=30, verbose=False)
    finally:
        bogp.GaussianProcess.fit = orig_fit
    assert len(minimum) == 5 and len(minimum[0]) == 2
    assert minimum[3] == 30  # function evaluations
    assert len(seen) == 21 and all(t is bogp.GaussianProcess for t in seen) and len(created) == 1
    assert minimum[1] < 1.0  # 30 evaluations of a 2-d sphere: the BO loop did optimise

    # warm starting (test_fmin.py:21-27)
    X = np.random.rand(10, 2) * 10 - 5
    y = [_sphere(x) for x in X]
    minimum = bayes_optim.fmin(_sphere, [-5] * 2, [5] * 2, x0=X, y0=y, max_FEs=20, verbose=False)
    assert minimum[2] == 20
    minimum = bayes_optim.fmin(_sphere, [-5] * 2, [5] * 2, x0=X, max_FEs=5, verbose=False)
    assert minimum[2] == 5


@pytest.mark.timeout(900)
def test_fmin_batch_mode_uses_the_fused_sweep(installed):
    """fmin(n_point=3) -> ParallelBO(MGFI) on the device classes; with the default-BFGS reroute the three proposals of an
    iteration come from one sweep."""
    bayes_optim, bogp, created = installed
    bogp.uninstall()
    bogp.install(bayes_optim, reroute_bfgs="sweep", sweep_budget=2000)
    minimum = bayes_optim.fmin(_sphere, [-5] * 2, [5] * 2, seed=1, max_FEs=19, n_point=3, verbose=False)
    assert minimum[3] == 19 and len(created) == 1 and created[0].uploads.count(2000) == 3  # 3 asks after the DoE, one sweep each


def _h(x):
    return np.sum(x) - 1


@pytest.mark.timeout(900)
def test_equality_constrained_BO_after_install(installed):
    """unittest/test_constraint.py:31-60 with the model built through the re-pointed name: BFGS + eq_fun -> the reference
    switches to OnePlusOne_Cholesky_CMA (base.py:208-209), which install() must hand to the reference's own optimiser."""
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, RealSpace
    from bayes_optim.surrogate import GaussianProcess

    dim = 2
    thetaL, thetaU = 1e-5 * np.ones(dim), np.ones(dim)
    np.random.seed(42)
    theta0 = np.random.rand(dim) * (thetaU - thetaL) + thetaL
    model = GaussianProcess(corr="squared_exponential", theta0=theta0, thetaL=thetaL, thetaU=thetaU, nugget=1e-1, random_state=42)
    assert type(model) is bogp.GaussianProcess and isinstance(model, GaussianProcess)
    opt = BO(search_space=RealSpace([0, 1]) * dim, obj_fun=lambda x: np.sum(np.array(x) ** 2) + 5 * np.sum(np.array(x)) + 10,
             eq_fun=_h, model=model, max_FEs=12, DoE_size=3, acquisition_fun="MGFI", acquisition_par={"t": 2},
             acquisition_optimization={"optimizer": "BFGS"}, verbose=False, random_seed=42)  # fmt: skip
    assert opt._optimizer == "OnePlusOne_Cholesky_CMA"
    xopt, _, __ = opt.run()
    assert opt.eval_count == 12 and np.isclose(_h(xopt), 0, atol=1e-1)


@pytest.mark.timeout(900)
def test_equality_constrained_sweep(installed):
    """The sweep with constraints: only sampled candidates the reference would accept (|h| <= 1e-1) compete."""
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, ParallelBO, RealSpace

    dim = 2
    f = lambda x: float(np.sum(np.array(x) ** 2) + 5 * np.sum(np.array(x)) + 10)  # noqa: E731
    gp = bogp.GaussianProcess(corr="squared_exponential", thetaL=1e-3 * np.ones(dim), thetaU=10 * np.ones(dim), nugget=1e-3, random_start=3)
    opt = BO(search_space=RealSpace([0, 1]) * dim, obj_fun=f, eq_fun=_h, model=gp, max_FEs=10, DoE_size=4, acquisition_fun="EI",
             acquisition_optimization={"optimizer": "sweep", "max_FEs": 3000}, verbose=False, random_seed=1)  # fmt: skip
    X = opt.ask()
    opt.tell(X, [f(x) for x in X])
    for _ in range(3):
        X = opt.ask()
        assert len(X) == 1 and abs(_h(X[0])) <= 1e-1
        opt.tell(X, [f(x) for x in X])
    gp2 = bogp.GaussianProcess(corr="squared_exponential", thetaL=1e-3 * np.ones(dim), thetaU=10 * np.ones(dim), nugget=1e-3, random_start=3)
    popt = ParallelBO(search_space=RealSpace([0, 1]) * dim, obj_fun=f, eq_fun=_h, model=gp2, max_FEs=20, DoE_size=5, n_point=3,
                      acquisition_fun="MGFI", acquisition_par={"t": 2}, acquisition_optimization={"optimizer": "sweep", "max_FEs": 3000},
                      verbose=False, random_seed=2)  # fmt: skip
    X = popt.ask()
    popt.tell(X, [f(x) for x in X])
    X = popt.ask()
    assert len(X) == 3 and all(abs(_h(x)) <= 1e-1 for x in X)


@pytest.mark.timeout(900)
def test_inequality_constraints_mixed_space_random_forest_after_install(installed):
    """unittest/test_constraint.py:63-90: mixed space + RandomForest + ineq_fun -> MIES on the reference's own MGFI; none of
    it is this package's business and all of it must still run."""
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, IntegerSpace, RealSpace
    from bayes_optim.surrogate import RandomForest

    space = (IntegerSpace([1, 10], var_name="mu") + IntegerSpace([1, 10], var_name="lambda") + RealSpace([0, 1], var_name="pc")
             + RealSpace([0.005, 0.5], var_name="p"))  # fmt: skip
    g = lambda x: [-x["pc"], x["mu"] - 1.9]  # noqa: E731
    opt = BO(search_space=space, obj_fun=lambda x: (x["pc"] - 0.2) ** 2 + x["mu"] + x["lambda"] + np.abs(x["p"] - 0.7), ineq_fun=g,
             model=RandomForest(levels=space.levels), max_FEs=8, DoE_size=3, eval_type="dict", acquisition_fun="MGFI",
             acquisition_par={"t": 2}, n_job=1, n_point=1, verbose=False, random_seed=42)  # fmt: skip
    assert opt._optimizer == "MIES"
    xopt, _, __ = opt.run()
    assert isinstance(xopt, dict) and all(np.array(g(xopt)) <= 0) and not created


@pytest.mark.timeout(900)
@pytest.mark.parametrize("inner", ["OnePlusOne_Cholesky_CMA", "MIES"])
def test_cma_and_mies_inner_optimisers_on_a_device_model(installed, inner):
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, RealSpace

    dim = 2
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(dim), corr="matern", thetaL=[1e-2] * dim, thetaU=[1e2] * dim, nugget=1e-6,
                              random_start=3, eval_budget=100)  # fmt: skip
    opt = BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(_sphere(x)), model=gp, DoE_size=5, max_FEs=9, verbose=False,
             n_point=1, acquisition_fun="EI", acquisition_optimization={"optimizer": inner, "max_FEs": 60, "n_restart": 2}, random_seed=5)  # fmt: skip
    opt.run()
    assert opt.eval_count == 9 and gp.is_fitted and type(_criterion_behind(opt._create_acquisition())).__module__.startswith("bogp")


@pytest.mark.timeout(900)
def test_reference_cpu_gp_as_model_after_install(installed):
    """A driver whose model is the reference's own CPU GaussianProcess gets the reference's own EI (the namespace
    dispatches per model) and its own BFGS loop -- the "model is not fitted yet" regression of r02."""
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, RealSpace
    from bayes_optim.surrogate.gaussian_process import GaussianProcess as CpuGP

    dim = 2
    new = lambda: CpuGP(mean=bayes_optim.trend.constant_trend(dim), corr="matern", thetaL=[1e-2] * dim, thetaU=[1e2] * dim,  # noqa: E731
                        nugget=1e-6, random_start=3, eval_budget=100)  # fmt: skip
    model = new()
    opt = BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(_sphere(x)), model=model, DoE_size=5, max_FEs=9,
             verbose=False, n_point=1, acquisition_fun="EI", random_seed=5)  # fmt: skip
    opt.run()
    crit = _criterion_behind(opt._create_acquisition())
    assert opt.eval_count == 9 and type(crit).__module__.startswith("bayes_optim") and not created
    import bayes_optim.base as rbase

    assert isinstance(crit, rbase.AcquisitionFunction.EI) and hasattr(rbase.AcquisitionFunction.EI, "plugin")
    assert not hasattr(rbase.AcquisitionFunction.UCB, "plugin") and rbase.AcquisitionFunction.norm is not None
    with pytest.raises(TypeError):  # a sweep was asked for BY NAME on a model that has no engine
        BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(_sphere(x)), model=new(), DoE_size=5, max_FEs=9, verbose=False,
           acquisition_fun="EI", acquisition_optimization={"optimizer": "sweep", "max_FEs": 100}, random_seed=5).run()  # fmt: skip


def test_unserved_gp_configuration_falls_back_to_the_cpu_class(installed):
    bayes_optim, bogp, created = installed
    from bayes_optim.surrogate.gaussian_process import GaussianProcess as CpuGP

    with pytest.warns(UserWarning, match="stays on the reference's CPU class"):
        gp = bayes_optim.GaussianProcess(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, optimizer="CMA")
    assert type(gp) is CpuGP and isinstance(gp, bayes_optim.GaussianProcess)
    with pytest.raises(ValueError):  # genuine argument errors surface as they would without install()
        bayes_optim.GaussianProcess(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 3)


def test_uninstall_restores_every_name(installed):
    bayes_optim, bogp, created = installed
    import bayes_optim.acquisition.acquisition_fun as racq
    import bayes_optim.acquisition.optim as roptim
    import bayes_optim.base as rbase
    import bayes_optim.bayes_opt as ropt
    from bayes_optim.surrogate.gaussian_process import GaussianProcess as CpuGP

    assert rbase.argmax_restart is not roptim.argmax_restart and bayes_optim.GaussianProcess is not CpuGP
    bogp.uninstall()
    assert rbase.argmax_restart is roptim.argmax_restart
    assert rbase.AcquisitionFunction is racq and ropt.AcquisitionFunction is racq
    assert bayes_optim.GaussianProcess is CpuGP and bayes_optim.surrogate.GaussianProcess is CpuGP
    bogp.install(bayes_optim, surrogate=False)
    assert bayes_optim.GaussianProcess is CpuGP


@pytest.mark.timeout(900)
@pytest.mark.parametrize("inner", ["sweep-BFGS", "sweep-device-BFGS"])
def test_sweep_bfgs_hybrid_under_the_real_driver(installed, inner):
    """The hybrid inner optimisers behind the reference's own `BO.ask()`: a sweep, then the polish of its top-k (on the oracle
    stand-in engine: the sequential L-BFGS-B route of `optim.polish_topk`; on the device: `bogp_polish`).  Every proposal is at
    least as good as the plain sweep's would have been from the same candidates."""
    bayes_optim, bogp, created = installed
    from bayes_optim import BO, RealSpace

    if inner == "sweep-device-BFGS":
        pytest.importorskip("bogp")  # the oracle stand-in has no device generator: the name must still be routed, and refuse clearly
    dim = 2
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(dim), corr="matern", thetaL=[1e-2] * dim, thetaU=[1e2] * dim, nugget=1e-6,
                              random_start=3, eval_budget=100)  # fmt: skip
    opt = BO(search_space=RealSpace([-5, 5]) * dim, obj_fun=lambda x: float(_sphere(x)), model=gp, DoE_size=6, max_FEs=9, verbose=False,
             n_point=1, acquisition_fun="EI", acquisition_optimization={"optimizer": inner, "max_FEs": 1500, "n_restart": 4}, random_seed=5)  # fmt: skip
    if inner == "sweep-device-BFGS" and not hasattr(created[0] if created else object(), "generate_candidates"):
        X = opt.ask()
        opt.tell(X, [float(_sphere(x)) for x in X])
        with pytest.raises(AttributeError):  # OracleEngine cannot draw candidates on a device it does not have
            opt.ask()
        return
    opt.run()
    assert opt.eval_count == 9 and gp.is_fitted
    crit = bogp.EI(model=gp, minimize=True)
    box = bogp.optim.Box([(-5.0, 5.0)] * dim, random_seed=1)
    x1, f1 = bogp.argmax_restart(crit, box, eval_budget=1500, optimizer="sweep")
    box = bogp.optim.Box([(-5.0, 5.0)] * dim, random_seed=1)
    x2, f2 = bogp.argmax_restart(crit, box, eval_budget=1500, n_restart=4, optimizer=inner)
    assert f2 >= f1 and len(x2) == dim


@pytest.mark.timeout(1500)
def test_reference_suite_passes_under_install(tmp_path):
    """The reference's OWN unittest files (BO / ParallelBO over every search-space flavour, pickling, fmin, constraints, warm
    data, surrogates, the inner optimisers), run UNMODIFIED in a subprocess whose pytest plugin calls `bogp.install()` before
    collection (tests/support/ref_suite_plugin.py; oracle-backed engine stand-in, no GPU here): all of them pass, as they do
    without the binding (59 / 59 in both modes when this was written; the plugin also papers over one scikit-learn keyword the
    reference predates, in both modes)."""
    import re
    import subprocess

    files = ["test_BO.py", "test_fmin.py", "test_constraint.py", "test_warmdata.py", "test_surrogate.py", "test_acq_optim.py",
             "test_mobo.py", "test_sampler.py", "test_Solution.py", "test_search_space.py"]
    # two tests need the real `py_expression_eval` (conditional search spaces): a stub in this image, failing with or without the binding
    skip = "not test_condition"  # test_search_space.py::test_condition, ::test_condition2
    env = dict(os.environ, BOGP_REF_SUITE_INSTALL="1",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), REF, os.path.join(ROOT, "oracle", "shims")]))
    cmd = [sys.executable, "-m", "pytest", "-p", "support.ref_suite_plugin", "-p", "no:cacheprovider", "-q", "--no-header", "-W", "ignore",
           "-n", "4"] + [os.path.join(REF, "unittest", f) for f in files] + ["-k", skip]
    res = subprocess.run(cmd + ["-rf"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1400)
    tail = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-500:]
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 55, res.stdout[-3000:] + res.stderr[-1000:]
    if res.returncode != 0:
        # The reference's tests are UNSEEDED optimisation runs (some assert on what a random search reaches): now and then one of
        # them fails, with or without the binding (observed once in five runs of this suite).  A failure counts only if the same
        # test fails again on its own, twice.
        failed = sorted(set(re.findall(r"^FAILED (\S+)", res.stdout, flags=re.M)))
        assert 0 < len(failed) <= 2, res.stdout[-3000:] + res.stderr[-1000:]
        for nodeid in failed:
            path, _, rest = nodeid.partition("::")  # (the summary names files relative to pytest's rootdir)
            nodeid = os.path.join(REF, "unittest", os.path.basename(path)) + "::" + rest
            again = [subprocess.run([sys.executable, "-m", "pytest", "-p", "support.ref_suite_plugin", "-p", "no:cacheprovider", "-q",
                                     "--no-header", "-W", "ignore", nodeid], cwd=str(tmp_path), env=env, capture_output=True,
                                    text=True, timeout=600) for _ in range(2)]
            assert any(r.returncode == 0 for r in again), "%s fails repeatedly under install():\n%s" % (nodeid, again[-1].stdout[-3000:])


@pytest.mark.timeout(900)
def test_install_restart_batch_reaches_the_models_fmin_builds(monkeypatch):
    """install(restart_batch=R): the GaussianProcess `fmin` constructs itself (`__init__.py:147-160`) fits with its MLE restarts through
    bogp_mle_batch (the engine stand-in runs the library's own L-BFGS-B on the oracle's likelihood); a configuration that stays on the
    reference's CPU class does not receive the keyword it would not know."""
    for p in (REF, os.path.join(ROOT, "oracle", "shims")):
        if p not in sys.path:
            sys.path.insert(0, p)
    warnings.filterwarnings("ignore")
    import bayes_optim

    import bogp
    from bayes_optim.surrogate.gaussian_process import GaussianProcess as CpuGP
    from support.oracle_engine import OracleEngine

    calls = []

    class Recording(OracleEngine):
        def mle_batch(self, *a, **k):
            calls.append(len(np.atleast_2d(a[2])))
            return super().mle_batch(*a, **k)

    monkeypatch.setattr(bogp._lib, "Engine", lambda device=0: Recording(device))
    undo = bogp.install(bayes_optim, restart_batch=3)
    try:
        gp = bayes_optim.GaussianProcess(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, random_start=5)
        assert type(gp) is bogp.GaussianProcess and gp.restart_batch == 3
        assert bayes_optim.GaussianProcess(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, restart_batch=0).restart_batch == 0
        with pytest.warns(UserWarning, match="stays on the reference's CPU class"):
            cpu = bayes_optim.GaussianProcess(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, optimizer="CMA")
        assert type(cpu) is CpuGP
        minimum = bayes_optim.fmin(_sphere, [-5] * 2, [5] * 2, seed=42, max_FEs=16, verbose=False)
        assert minimum[3] == 16 and calls and max(calls) <= 3
        assert minimum[1] < 5.0
    finally:
        undo()
    assert bayes_optim.GaussianProcess is not None and not bogp.integration._SURROGATE_DEFAULTS


def test_subclasses_and_foreign_trends_of_the_dispatching_class_stay_on_the_host(installed):
    """(1) `class My(bayes_optim.GaussianProcess)` inherits the dispatching metaclass -- its overrides were written against the
    reference's class, so it is built on the host class with the overrides in place, not replaced by a plain device object; (2) a trend
    basis the device does not evaluate is found at CONSTRUCTION (fallback + warning), not at fit()."""
    bayes_optim, bogp, created = installed
    from bayes_optim.surrogate.gaussian_process import GaussianProcess as CpuGP
    from bayes_optim.surrogate.gaussian_process.trend import BasisExpansionTrend

    class My(bayes_optim.GaussianProcess):
        def predict(self, X, eval_MSE=False, batch_size=None):
            return "overridden"

    m = My(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2)
    assert isinstance(m, CpuGP) and not isinstance(m, bogp.GaussianProcess) and m.predict(None) == "overridden"

    # the usual pattern -- zero-argument super() in __init__ and in an overriding fit -- must resolve; the realised class is
    # built once (type identity, isinstance against the user's name), further subclassing works, and instances pickle
    global _Sup, _Sub  # (module level names: pickle looks classes up by module attribute)

    class _Sup(bayes_optim.GaussianProcess):
        def __init__(self, *a, tag="t", **kw):
            super().__init__(*a, **kw)
            self.tag = tag

        def fit(self, X, y):
            self.fits = getattr(self, "fits", 0) + 1
            return super().fit(X, y)

        @property
        def label(self):
            return "sup:" + super().__class__.__name__

    class _Sub(_Sup):
        def fit(self, X, y):
            self.sub_fits = getattr(self, "sub_fits", 0) + 1
            return super().fit(X, y)

    a = _Sup(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, nugget=1e-6, random_start=2, tag="a")
    b = _Sup(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, nugget=1e-6, random_start=2)
    assert type(a) is type(b) and isinstance(a, _Sup) and isinstance(a, CpuGP) and issubclass(_Sub, _Sup) and not isinstance(m, _Sup)
    assert a.tag == "a" and b.tag == "t" and a.label.startswith("sup:")
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (12, 2))
    y = np.sum(X**2, axis=1).reshape(-1, 1)
    np.random.seed(1)
    assert a.fit(X, y) is a and a.fits == 1 and a.is_fitted
    c = _Sub(corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2, nugget=1e-6, random_start=2)
    assert isinstance(c, _Sub) and isinstance(c, _Sup) and not isinstance(a, _Sub)
    np.random.seed(1)
    c.fit(X, y)
    assert c.sub_fits == 1 and c.fits == 1
    np.testing.assert_allclose(c.predict(X[:3]), a.predict(X[:3]))
    import pickle

    a2 = pickle.loads(pickle.dumps(a))
    assert type(a2) is type(a) and a2.tag == "a" and a2.fits == 1
    np.testing.assert_array_equal(a2.predict(X[:3]), a.predict(X[:3]))

    class CubicTrend(BasisExpansionTrend):
        def __init__(self, n_feature, beta=None):
            super().__init__(n_feature, n_feature + 1, beta)

        def F(self, X):
            X = self.check_input(X)
            return np.c_[np.ones(len(X)), X**3]

    with pytest.warns(UserWarning, match="stays on the reference's CPU class"):
        gp = bayes_optim.GaussianProcess(mean=CubicTrend(2), corr="matern", thetaL=[1e-2] * 2, thetaU=[1e2] * 2)
    assert type(gp) is CpuGP
