"""Generate golden vectors by IMPORTING the reference (`/root/reference/bayes_optim`) in the build container.

Run:  python oracle/make_golden.py            (writes tests/golden/G*.npz)

The reference cannot travel to the GPU box, so its outputs on seeded inputs are committed as data.
Each .npz holds inputs + the reference's own outputs; nothing here is reference source.  `oracle/shims/`
provides stand-ins for three third-party packages missing from this image (pyDOE, sobol_seq,
py_expression_eval) so that `import bayes_optim` succeeds.

State pinning follows SURVEY.md Appendix A: `_check_data` -> one `log_likelihood_concentrated(par, env)`
call -> copy env as `fit` does (gpr.py:402-415) -> `compute_beta_gamma`.
"""
import functools
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402

import bayes_optim  # noqa: E402,F401
from bayes_optim.acquisition import acquisition_fun as AF  # noqa: E402
from bayes_optim.surrogate import GaussianProcess, trend  # noqa: E402
from bayes_optim.surrogate.gaussian_process.kernel import matern  # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
VERS = dict(numpy=np.__version__, scipy=scipy.__version__)


def make_data(seed, N, d, lo=-5.0, hi=5.0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(lo, hi, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = (y - y.mean()) / y.std()  # what BaseBO.update_model does (base.py:437-441)
    return X, y.reshape(-1, 1)


def pin(gp, X, y, par):
    gp._check_data(X, y)
    env = {}
    llf = gp.log_likelihood_concentrated(np.asarray(par, float), env)
    assert np.isfinite(llf), llf
    n_theta = len(gp.thetaL)
    gp.theta_ = np.asarray(par[:n_theta], float)
    gp.noise_var = env["noise_var"]
    gp.sigma2 = np.atleast_1d(env["sigma2"]).astype(float)
    gp.rho, gp.Yt, gp.C = env["rho"], env["Yt"], env["C"]
    if gp.estimate_trend:
        gp.Ft, gp.G, gp.Q = env["Ft"], env["G"], env["Q"]
    gp.compute_beta_gamma()
    gp.is_fitted = True
    return llf


def state_dict(gp, llf):
    d = dict(
        X=gp.X, y=gp.y, theta=gp.theta_, sigma2=gp.sigma2, noise_var=np.atleast_1d(gp.noise_var).astype(float),
        C=gp.C, gamma=gp.gamma, rho=gp.rho, Yt=gp.Yt, beta=np.asarray(gp.mean.beta, float), llf=float(llf),
        estimate_trend=bool(gp.estimate_trend),
    )  # fmt: skip
    if gp.estimate_trend:
        d.update(Ft=gp.Ft, G=gp.G, Q=gp.Q)
    return d


def acq_rows(gp, Xs, minimize=True, plugin=None, ts=(1.0, 2.0, 100.0), alphas=(0.5,), eps=(1e-10,)):
    """Row-by-row acquisition values exactly as the reference returns them for a (1,d) input."""
    out = {}

    def rows(c):
        v = np.empty(len(Xs))
        for i, x in enumerate(Xs):
            r = c(x.reshape(1, -1))
            v[i] = float(np.asarray(r, dtype=float).ravel()[0])
        return v

    kw = dict(model=gp, minimize=minimize)
    pkw = dict(kw, plugin=plugin)
    out["EI"] = rows(AF.EI(**pkw))
    for e in eps:
        out["EpsilonPI_%g" % e] = rows(AF.EpsilonPI(epsilon=e, **pkw))
    for a in alphas:
        out["UCB_%g" % a] = rows(AF.UCB(alpha=a, **kw))
    for t in ts:
        out["MGFI_%g" % t] = rows(AF.MGFI(t=t, **pkw))
    out["plugin_eff"] = np.array([AF.EI(**pkw).plugin])
    for k in list(out):
        if k != "plugin_eff":
            out["argmax_" + k] = np.array([int(np.argmax(out[k]))])
    return out


def grad_rows(gp, Xg, minimize=True, plugin=None, with_mgfi=True):
    out = {}
    dmu, dmse = [], []
    for x in Xg:
        a, b = gp.gradient(x)
        dmu.append(a)
        dmse.append(b)
    out["grad_mu"] = np.array(dmu)  # (n, d, n_t)
    out["grad_mse"] = np.array(dmse)  # (n, d, 1)
    crit = {
        "EI": AF.EI(model=gp, minimize=minimize, plugin=plugin),
        "EpsilonPI": AF.EpsilonPI(model=gp, minimize=minimize, plugin=plugin),
        "UCB": AF.UCB(model=gp, minimize=minimize),
    }
    if with_mgfi:
        crit["MGFI_2"] = AF.MGFI(t=2.0, model=gp, minimize=minimize, plugin=plugin)
    for name, c in crit.items():
        vals, dxs = [], []
        for x in Xg:
            v, dx = c(x.reshape(1, -1), return_dx=True)
            vals.append(float(np.asarray(v, float).ravel()[0]))
            dxs.append(np.asarray(dx, float).ravel())
        out["dx_val_" + name] = np.array(vals)
        out["dx_" + name] = np.array(dxs)
    return out


def save(name, **kw):
    kw.update({"ver_" + k: np.array(v) for k, v in VERS.items()})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **kw)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def llf_table(gp, pars):
    vals, grads = [], []
    for p in pars:
        v, g = gp.log_likelihood_concentrated(np.asarray(p, float), eval_grad=True)
        vals.append(float(v))
        grads.append(np.asarray(g, float).ravel())
    return np.array(vals), np.array(grads)


def main():
    # ---- G1: SE + simple kriging (beta=0) + nugget 1e-6 ("noisy"), N=64, d=5 --------------------
    X, y = make_data(1, 64, 5)
    d = 5
    gp = GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    par = np.r_[np.full(d, 0.05) * np.linspace(0.6, 1.5, d), 0.9]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(101)
    Xs = rng.uniform(-5, 5, size=(256, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    Xg = Xs[:8]
    save("G1_se_sk_noisy", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(0), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **grad_rows(gp, Xg))  # fmt: skip

    # ---- G2: fmin's model: Matern-3/2 + constant_trend(beta=None) (ordinary kriging), N=50, d=2 ---
    X, y = make_data(2, 50, 2)
    d = 2
    gp = GaussianProcess(mean=trend.constant_trend(d), corr="matern", thetaL=[1e-2] * d, thetaU=[1e4] * d, nugget=1e-6)
    par = np.r_[0.3, 0.41, 0.8]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(102)
    Xs = rng.uniform(-5, 5, size=(256, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    save("G2_m32_ok_noisy", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(2), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **grad_rows(gp, Xs[:8]))  # fmt: skip

    # ---- G3: Matern-5/2 via partial(matern, nu=2.5); values only (cannot be fitted/differentiated) --
    X, y = make_data(3, 128, 20)
    d = 20
    gp = GaussianProcess(corr=functools.partial(matern, nu=2.5), thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    par = np.r_[np.full(d, 0.01) * np.linspace(0.7, 1.4, d), 0.9]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(103)
    Xs = rng.uniform(-5, 5, size=(256, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    save("G3_m52_sk_noisy", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(3), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs))  # fmt: skip

    # ---- G4: nugget=0 "noiseless" SE with estimated constant trend (sigma2 from rho; matrix_rank path) ----
    X, y = make_data(4, 48, 3)
    d = 3
    gp = GaussianProcess(mean=trend.constant_trend(d), corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=0)
    par = np.array([0.30, 0.22, 0.41])
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(104)
    Xs = rng.uniform(-5, 5, size=(128, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    save("G4_se_ok_noiseless", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(0), mode=np.array(0),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **grad_rows(gp, Xs[:4]))  # fmt: skip

    # ---- G5: noise_estim=True, SE, simple kriging ------------------------------------------------
    X, y = make_data(5, 40, 4)
    d = 4
    gp = GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6, noise_estim=True)
    par = np.r_[0.06, 0.04, 0.09, 0.05, 0.97]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(105)
    Xs = rng.uniform(-5, 5, size=(128, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    save("G5_se_sk_noise_estim", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(0), mode=np.array(2),
         **state_dict(gp, llf), **acq_rows(gp, Xs))  # fmt: skip

    # ---- G6: llf + gradient tables, three modes x {SE, Matern-3/2}, 5 parameter vectors each ------
    X, y = make_data(6, 40, 3)
    d = 3
    rng = np.random.default_rng(106)
    tabs = {}
    for kname, kid in (("squared_exponential", 0), ("matern", 2)):
        for mname, mid, kw, extra in (
            ("noiseless", 0, dict(nugget=0), 0),
            ("noisy", 1, dict(nugget=1e-6), 1),
            ("noise_estim", 2, dict(nugget=1e-6, noise_estim=True), 1),
        ):
            for tname, mean in (("sk", None), ("ok", trend.constant_trend(d))):
                gp = GaussianProcess(mean=mean, corr=kname, thetaL=[1e-4] * d, thetaU=[1e2] * d, **kw)
                gp._check_data(X, y)
                pars = []
                for _ in range(5):
                    th = 10 ** rng.uniform(-1.5, -0.3, size=d)
                    if mid == 0:
                        pars.append(th)
                    elif mid == 1:
                        pars.append(np.r_[th, rng.uniform(0.3, 1.2)])
                    else:
                        pars.append(np.r_[th, rng.uniform(0.7, 0.999)])
                v, g = llf_table(gp, pars)
                key = "k%d_m%d_%s" % (kid, mid, tname)
                tabs[key + "_par"] = np.array(pars)
                tabs[key + "_llf"] = v
                tabs[key + "_grad"] = g
    save("G6_llf_tables", X=X, y=y, noise_var=np.array([1e-6]), **tabs)

    # ---- G7: edge rows (noiseless SE: candidate == training point -> MSE clipped to 0), far point,
    #          minimize=False ------------------------------------------------------------------------
    X, y = make_data(7, 24, 2)
    d = 2
    gp = GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=0)
    par = np.array([0.5, 0.4])
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(107)
    Xs = np.r_[X[:6], np.array([[40.0, -40.0], [1e3, 1e3]]), rng.uniform(-5, 5, size=(24, d))]
    mu, mse = gp.predict(Xs, eval_MSE=True)
    a_min = acq_rows(gp, Xs)
    a_max = {"max_" + k: v for k, v in acq_rows(gp, Xs, minimize=False).items()}
    a_plg = {"plg_" + k: v for k, v in acq_rows(gp, Xs, plugin=-0.3).items()}
    save("G7_edges", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(0), mode=np.array(0),
         **state_dict(gp, llf), **a_min, **a_max, **a_plg)  # fmt: skip

    # ---- G8: mid-size N=512, d=10, 2048 candidates (mu, MSE, EI only); state regenerated from par ---
    X, y = make_data(8, 512, 10)
    d = 10
    gp = GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    par = np.r_[np.full(d, 0.02), 0.9]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(108)
    Xs = rng.uniform(-5, 5, size=(2048, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    c = AF.EI(model=gp, minimize=True)
    eiv = np.array([float(np.asarray(c(x.reshape(1, -1)), float).ravel()[0]) for x in Xs])
    # C (2 MB) is not stored: the test rebuilds the state from (X, y, par) with the oracle and checks gamma
    save("G8_mid", par=par, Xs=Xs.astype(np.float32).astype(np.float64), note=np.array("Xs stored after f32 round-trip"),
         seed=np.array(8), N=np.array(512), d=np.array(10), llf=np.array(llf),
         gamma=gp.gamma, sigma2=gp.sigma2, kernel=np.array(0), mode=np.array(1),
         **_g8_outputs(gp, Xs.astype(np.float32).astype(np.float64)))  # fmt: skip

    # ---- G12: absolute_exponential (the third kernel the reference can actually fit): state, values, gradients,
    #           llf tables in the three modes -----------------------------------------------------------------------
    X, y = make_data(12, 60, 4)
    d = 4
    gp = GaussianProcess(mean=trend.constant_trend(d), corr="absolute_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    par = np.r_[0.11, 0.07, 0.16, 0.09, 0.85]
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(112)
    Xs = rng.uniform(-5, 5, size=(256, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    tabs = {}
    rng2 = np.random.default_rng(212)
    for mname, mid, kw in (("noiseless", 0, dict(nugget=0)), ("noisy", 1, dict(nugget=1e-6)), ("noise_estim", 2, dict(nugget=1e-6, noise_estim=True))):
        for tname, mean in (("sk", None), ("ok", trend.constant_trend(d))):
            g2 = GaussianProcess(mean=mean, corr="absolute_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, **kw)
            g2._check_data(X, y)
            pars = []
            for _ in range(4):
                th = 10 ** rng2.uniform(-1.3, -0.5, size=d)
                pars.append(th if mid == 0 else np.r_[th, rng2.uniform(0.4, 1.1) if mid == 1 else rng2.uniform(0.7, 0.999)])
            v, gr = llf_table(g2, pars)
            key = "t_m%d_%s" % (mid, tname)
            tabs[key + "_par"], tabs[key + "_llf"], tabs[key + "_grad"] = np.array(pars), v, gr
    save("G12_absexp_ok_noisy", par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(4), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **grad_rows(gp, Xs[:8]), **tabs)  # fmt: skip

    # ---- G9: plumbing invariants of fmin (trajectory depends on the LHS stand-in; not a golden) -----
    np.random.seed(42)
    res = bayes_optim.fmin(lambda x: float(np.sum(np.asarray(x) ** 2)), [-5.0, -5.0], [5.0, 5.0], max_FEs=30, seed=42, verbose=False)
    save("G9_fmin_plumbing", n_ret=np.array(len(res)), n_x=np.array(len(res[0])), n_iter=np.array(res[2]),
         n_eval=np.array(res[3]))  # fmt: skip
    golden_fit()


def golden_fit():
    """G10: full GaussianProcess.fit (MLE through L-BFGS-B restarts) by the reference on seeded data; the GPU
    class replays the same host loop (same global np.random stream) and must land on the same optimum."""
    out = {}
    for tag, kw, d, N in (
        ("se_sk_noisy", dict(corr="squared_exponential", nugget=1e-6), 3, 40),
        ("m32_ok_noisy", dict(corr="matern", nugget=1e-6, mean="ok"), 2, 30),
        ("se_sk_noise_estim", dict(corr="squared_exponential", nugget=1e-6, noise_estim=True), 3, 40),
    ):
        X, y = make_data(10 + d + N, N, d)
        y = y + 0.05 * np.random.default_rng(5).standard_normal(y.shape)  # a little noise keeps llf <= 0
        kw = dict(kw)
        mean = trend.constant_trend(d) if kw.pop("mean", None) == "ok" else None
        gp = GaussianProcess(mean=mean, thetaL=[1e-3] * d, thetaU=[1e2] * d, optimizer="BFGS", wait_iter=3,
                             random_start=5, eval_budget=100 * d, **kw)  # fmt: skip
        traj = []
        if tag == "se_sk_noisy":  # G11: every (par, llf, grad) the reference's MLE visits
            orig = gp.log_likelihood_concentrated

            def rec(par, env=None, eval_grad=False, _orig=orig, _traj=traj):
                out = _orig(par, env, eval_grad)
                if eval_grad:
                    _traj.append((np.array(par, float), float(out[0]), np.asarray(out[1], float).ravel()))
                return out

            gp.log_likelihood_concentrated = rec
        np.random.seed(123)
        gp.fit(X, y)
        if traj:
            save("G11_mle_trajectory", X=X, y=y, par=np.array([t[0] for t in traj]), llf=np.array([t[1] for t in traj]),
                 grad=np.array([t[2] for t in traj]), kernel=np.array(0), mode=np.array(1), noise_var=np.array([1e-6]))  # fmt: skip
        Xs = np.random.default_rng(6).uniform(-5, 5, size=(64, d))
        mu, mse = gp.predict(Xs, eval_MSE=True)
        out.update({
            tag + "_X": X, tag + "_y": y, tag + "_theta": gp.theta_, tag + "_sigma2": gp.sigma2,
            tag + "_noise_var": np.atleast_1d(gp.noise_var).astype(float), tag + "_llf": np.array(gp.log_likelihood_),
            tag + "_Xs": Xs, tag + "_mu": mu, tag + "_mse": mse, tag + "_beta": np.asarray(gp.mean.beta, float),
        })  # fmt: skip
    save("G10_fit", **out)


def _g8_outputs(gp, Xs):
    mu, mse = gp.predict(Xs, eval_MSE=True)
    c = AF.EI(model=gp, minimize=True)
    eiv = np.array([float(np.asarray(c(x.reshape(1, -1)), float).ravel()[0]) for x in Xs])
    return dict(mu=mu, mse=mse, EI=eiv, argmax_EI=np.array([int(np.argmax(eiv))]))


def golden_trends():
    """G13-G15: polynomial trend bases with p > 1 (trend.py:94-142) through the reference's fit-side and predict-side
    code: pinned state, predictor, acquisition rows, input gradients (linear only: quadratic_trend.Jacobian raises),
    likelihood + gradient tables in the three estimation modes."""
    cases = (
        # name, trend class, trend id, kernel name, kernel id, N, d, fixed beta
        ("G13_linear_uk_se", trend.linear_trend, 1, "squared_exponential", 0, 48, 3, None),
        ("G14_quadratic_uk_m32", trend.quadratic_trend, 2, "matern", 2, 64, 3, None),
        ("G15_linear_sk_se", trend.linear_trend, 1, "squared_exponential", 0, 40, 4, [0.1, -0.05, 0.02, 0.03, -0.01]),
    )
    for seed, (name, tcls, tid, corr, kid, N, d, beta) in enumerate(cases, start=13):
        X, y = make_data(seed, N, d)
        # a linear component for the trend to pick up + noise (a quadratic basis would otherwise fit sum(x^2) exactly
        # and every likelihood would be > 0, i.e. rejected, gpr.py:981-982)
        y = y + 0.3 * X[:, :1] / 5.0 + 0.25 * np.random.default_rng(300 + seed).standard_normal(y.shape)
        gp = GaussianProcess(mean=tcls(d, beta=beta), corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
        par = np.r_[np.full(d, 0.06) * np.linspace(0.7, 1.4, d), 0.85]
        llf = pin(gp, X, y, par)
        rng = np.random.default_rng(100 + seed)
        Xs = rng.uniform(-5, 5, size=(192, d))
        mu, mse = gp.predict(Xs, eval_MSE=True)
        extra = grad_rows(gp, Xs[:8]) if tid == 1 else {}
        tabs = {}
        rng2 = np.random.default_rng(200 + seed)
        for mname, mid, kw in (("noiseless", 0, dict(nugget=0)), ("noisy", 1, dict(nugget=1e-6)), ("noise_estim", 2, dict(nugget=1e-6, noise_estim=True))):
            g2 = GaussianProcess(mean=tcls(d, beta=beta), corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d, **kw)
            g2._check_data(X, y)
            pars = []
            for _ in range(4):
                th = 10 ** rng2.uniform(-1.6, -0.8, size=d)
                pars.append(th if mid == 0 else np.r_[th, rng2.uniform(0.4, 1.1) if mid == 1 else rng2.uniform(0.7, 0.999)])
            v, gr = llf_table(g2, pars)
            key = "t_m%d" % mid
            tabs[key + "_par"], tabs[key + "_llf"], tabs[key + "_grad"] = np.array(pars), v, gr
        save(name, par=par, Xs=Xs, mu=mu, mse=mse, kernel=np.array(kid), mode=np.array(1), trend=np.array(tid),
             **state_dict(gp, llf), **acq_rows(gp, Xs), **extra, **tabs)  # fmt: skip


def golden_reml():
    """G16: the restricted likelihood (gpr.py:813-918) and its gradient as the reference computes them, three modes x
    {simple, ordinary kriging} x {SE, Matern-3/2}.  Only the FUNCTION is a golden: `fit(likelihood="restricted")` itself
    raises TypeError in the reference (gpr.py:405, sigma2 comes back as a scalar)."""
    X, y = make_data(16, 40, 3)
    y = y + 0.2 * np.random.default_rng(316).standard_normal(y.shape)
    d = 3
    out = dict(X=X, y=y)
    rng = np.random.default_rng(216)
    n = 0
    for kid, corr in ((0, "squared_exponential"), (2, "matern")):
        for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
            for tname, mean in (("sk", None), ("ok", trend.constant_trend(d))):
                gp = GaussianProcess(mean=mean, corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d, likelihood="restricted", **kw)
                gp._check_data(X, y)
                pars, vals, grads = [], [], []
                for _ in range(4):
                    th = 10 ** rng.uniform(-1.5, -0.6, size=d)
                    p = np.r_[th, rng.uniform(0.3, 1.2)]
                    if mid == 2:
                        p = np.r_[p, 10 ** rng.uniform(-4, -1)]
                    v, g = gp.log_likelihood_restricted(p, eval_grad=True)
                    pars.append(p)
                    vals.append(float(v))
                    grads.append(np.asarray(g, float).ravel())
                    n += 1
                key = "k%d_m%d_%s" % (kid, mid, tname)
                out[key + "_par"], out[key + "_llf"], out[key + "_grad"] = np.array(pars), np.array(vals), np.array(grads)
    assert n == 48
    save("G16_reml_tables", **out)


def golden_multitarget():
    """G17: y with three columns (gpr.py:463, 490, 502-505, 931-1040).  In the reference only a FIXED constant trend
    survives `fit` with several targets (estimating it raises at :787), so that is the golden: likelihood + gradient
    tables for the three modes x {SE, Matern-3/2}, one pinned state per mode with predictions, and one real `fit`."""
    rng = np.random.default_rng(417)
    N, d = 48, 4
    X = rng.uniform(-5, 5, size=(N, d))
    Y = np.c_[np.sum(X**2, 1), np.sum(np.sin(X), 1), X[:, 0] * X[:, 1] - X[:, 2]]
    Y = (Y - Y.mean(0)) / Y.std(0) + 0.25 * rng.standard_normal((N, 3)) * np.array([1.0, 0.6, 1.4])
    Xs = rng.uniform(-5, 5, size=(200, d))
    out = dict(X=X, y=Y, Xs=Xs)
    n = 0
    for kid, corr in ((0, "squared_exponential"), (2, "matern")):
        for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-3)), (2, dict(nugget=1e-6, noise_estim=True))):
            gp = GaussianProcess(mean=trend.constant_trend(d, beta=0.0), corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d, **kw)
            gp._check_data(X, Y)
            pars = []
            for _ in range(4):
                th = 10 ** rng.uniform(-1.6, -0.7, size=d)
                pars.append(th if mid == 0 else np.r_[th, rng.uniform(0.4, 0.95)])
            vals, grads = llf_table(gp, pars)
            assert np.all(np.isfinite(vals)), (kid, mid, vals)
            key = "k%d_m%d" % (kid, mid)
            out[key + "_par"], out[key + "_llf"], out[key + "_grad"] = np.array(pars), vals, grads
            n += len(pars)
            llf = pin(gp, X, Y, pars[0])
            mu, mse = gp.predict(Xs, eval_MSE=True)
            out.update({key + "_st_" + k: v for k, v in state_dict(gp, llf).items() if k not in ("X", "y", "C")})
            out[key + "_mu"], out[key + "_mse"] = mu, mse
    assert n == 24
    np.random.seed(7)
    gp = GaussianProcess(mean=trend.constant_trend(d, beta=0.0), corr="squared_exponential", thetaL=[1e-3] * d, thetaU=[10.0] * d,
                         nugget=1e-3, optimizer="BFGS", wait_iter=3, random_start=3, eval_budget=300, random_state=7)  # fmt: skip
    gp.fit(X, Y)
    mu, mse = gp.predict(Xs, eval_MSE=True)
    out.update(fit_theta=gp.theta_, fit_sigma2=gp.sigma2, fit_mu=mu, fit_mse=mse, fit_gamma=gp.gamma)
    save("G17_multitarget", **out)


def golden_isotropic():
    """G18: one theta for all d > 1 dimensions (kernel.py:319-320).  The likelihood value is the ordinary one; its gradient
    is what the reference's loops produce when they index the (N, N, d) tensor of `corr_grad_theta` by PARAMETER
    (gpr.py:1001-1037): the derivative along dimension 0 in the theta row and, in the noisy mode, along dimension 1 in
    the sigma2 row.  That vector drives the reference's MLE, so it is the golden."""
    X, y = make_data(18, 44, 3)
    y = y + 0.15 * np.random.default_rng(518).standard_normal(y.shape)
    d = 3
    out = dict(X=X, y=y)
    rng = np.random.default_rng(618)
    n = 0
    for kid, corr in ((0, "squared_exponential"), (2, "matern"), (4, "absolute_exponential")):
        for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
            for tname, mean in (("sk", trend.constant_trend(d, beta=0.0)), ("ok", trend.constant_trend(d))):
                gp = GaussianProcess(mean=mean, corr=corr, thetaL=[1e-4], thetaU=[1e2], **kw)
                gp._check_data(X, y)
                pars = []
                for _ in range(3):
                    th = 10 ** rng.uniform(-1.8, -0.9, size=1)
                    pars.append(th if mid == 0 else np.r_[th, rng.uniform(0.4, 0.95)])
                vals, grads = llf_table(gp, pars)
                key = "k%d_m%d_%s" % (kid, mid, tname)
                out[key + "_par"], out[key + "_llf"], out[key + "_grad"] = np.array(pars), vals, grads
                n += int(np.isfinite(vals).sum())
    assert n >= 40, n
    save("G18_isotropic_tables", **out)


def golden_hessian_prior():
    """G19: GaussianProcess.Hessian (gpr.py:578-598; squared exponential, the only kernel corr_Hessian defines) and
    prior_cov (gpr.py:318-353) as the reference returns them, for a simple-kriging and an ordinary-kriging SE model."""
    out = {}
    for tag, mean_of, nug in (("sk", lambda d: None, 1e-6), ("ok", lambda d: trend.constant_trend(d), 0)):
        X, y = make_data(19, 60, 4)
        d = 4
        gp = GaussianProcess(mean=mean_of(d), corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=nug)
        par = np.r_[[0.03, 0.05, 0.02, 0.08], 0.9] if nug else np.array([0.03, 0.05, 0.02, 0.08])
        pin(gp, X, y, par)
        rng = np.random.default_rng(719)
        P = rng.uniform(-5, 5, size=(6, d))
        out.update({tag + "_X": X, tag + "_y": y, tag + "_par": par, tag + "_P": P,
                    tag + "_H": np.array([gp.Hessian(p) for p in P]),
                    tag + "_corr": gp.prior_cov(P, corr=True), tag + "_cov": gp.prior_cov(P)})  # fmt: skip
    save("G19_hessian_prior_cov", **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["trends"]:
        golden_trends()
    elif sys.argv[1:] == ["reml"]:
        golden_reml()
    elif sys.argv[1:] == ["multitarget"]:
        golden_multitarget()
    elif sys.argv[1:] == ["isotropic"]:
        golden_isotropic()
    elif sys.argv[1:] == ["hessian"]:
        golden_hessian_prior()
    else:
        main()
        golden_trends()
        golden_reml()
        golden_multitarget()
        golden_isotropic()
        golden_hessian_prior()
