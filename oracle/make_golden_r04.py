"""Round-4 goldens made by IMPORTING the reference (build container only):  python oracle/make_golden_r04.py

G31_matern_nu08_ok_noisy / G32_matern_nu37_sk_noisy: the general-nu arm of the reference's Matern kernel (kernel.py:201-207:
scipy.special.kv), reachable through corr=functools.partial(matern, nu=...) for an order outside {1/2, 3/2, 5/2}.  Like cubic and
generalized_exponential the reference can evaluate it but not differentiate it (corr_grad_theta / corr_dx define nothing for a callable
corr), so: a pinned state, the posterior, the criteria row by row, np.argmax, and tables of likelihood VALUES in
the three estimation modes.  The stored `par` has the engine's layout [theta_1 .. theta_d, nu, sigma2] (include/bogp.h: the order
travels as the last theta entry); the reference takes nu as a keyword.

G33_reml_multitarget: see golden_reml_multitarget."""
import functools
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402

from bayes_optim.surrogate import GaussianProcess, trend  # noqa: E402
from bayes_optim.surrogate.gaussian_process.kernel import matern  # noqa: E402

from oracle.make_golden import acq_rows, make_data, pin, save, state_dict  # noqa: E402

warnings.filterwarnings("ignore")


def one(name, nu, d, ok, par_ref, seed):
    corr = functools.partial(matern, nu=nu)
    X, y = make_data(seed, 70, d)
    y = y + 0.05 * np.random.default_rng(seed + 1).standard_normal(y.shape)
    mean = (lambda: trend.constant_trend(d)) if ok else (lambda: trend.constant_trend(d, beta=0))
    gp = GaussianProcess(mean=mean(), corr=corr, thetaL=[1e-5] * d, thetaU=[1e2] * d, nugget=1e-6)
    llf = pin(gp, X, y, par_ref)
    rng = np.random.default_rng(seed + 2)
    Xs = rng.uniform(-5, 5, size=(256, d))
    Xs[5] = X[9]  # a candidate on a training point: dists == 0 -> eps (kernel.py:203)
    mu, mse = gp.predict(Xs, eval_MSE=True)
    tabs = {}
    rng2 = np.random.default_rng(seed + 3)
    for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
        for tname, est in (("sk", False), ("ok", True)):
            g2 = GaussianProcess(mean=trend.constant_trend(d) if est else trend.constant_trend(d, beta=0), corr=corr,
                                 thetaL=[1e-5] * d, thetaU=[1e2] * d, **kw)  # fmt: skip
            g2._check_data(X, y)
            pars, vals = [], []
            for _ in range(4):
                th = 10 ** rng2.uniform(-1.5, -0.4, size=d) * (8.0 if mid == 0 else 1.0)
                pr = th if mid == 0 else np.r_[th, rng2.uniform(0.4, 1.1) if mid == 1 else rng2.uniform(0.7, 0.999)]
                vals.append(float(g2.log_likelihood_concentrated(np.asarray(pr, float))))
                pars.append(np.r_[pr[:d], nu, pr[d:]])  # the engine's layout
            key = "t_m%d_%s" % (mid, tname)
            tabs[key + "_par"], tabs[key + "_llf"] = np.array(pars), np.array(vals)
    par_engine = np.r_[par_ref[:d], nu, par_ref[d:]]
    save(name, par=par_engine, nu=np.array(nu), Xs=Xs, mu=mu, mse=mse, kernel=np.array(7), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **tabs)  # fmt: skip


def golden_reml_multitarget():
    """G33: the restricted likelihood with SEVERAL targets (gpr.py:813-918 called on a model whose y has 2 / 3 columns).  The reference
    yields a value -- its scalar terms broadcast over the n_t x n_t matrix rho^T rho and everything is summed (:861-866) -- and raises
    ValueError in the gradient (a (1, N n_t) by (N, N) product, :875, :896); like every multi-target model it needs a FIXED constant trend."""
    d = 3
    X, y1 = make_data(33, 45, d)
    rng = np.random.default_rng(333)
    Y = np.column_stack([y1.ravel(), np.cos(X).sum(axis=1) + 0.1 * rng.standard_normal(len(X)), X[:, 0] * X[:, 1] - 0.5 * X[:, 2]])
    out = dict(X=X, Y=Y, beta=np.array(0.15))
    n = 0
    for T in (2, 3):
        for kid, corr in ((0, "squared_exponential"), (2, "matern")):
            for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-4)), (2, dict(nugget=1e-6, noise_estim=True))):
                gp = GaussianProcess(mean=trend.constant_trend(d, beta=0.15), corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d,
                                     likelihood="restricted", **kw)  # fmt: skip
                gp._check_data(X, Y[:, :T])
                gp.X, gp.y = X, Y[:, :T]
                pars, vals = [], []
                for _ in range(3):
                    th = 10 ** rng.uniform(-1.5, -0.6, size=d)
                    p = np.r_[th, rng.uniform(0.3, 1.2)]
                    if mid == 2:
                        p = np.r_[p, 10 ** rng.uniform(-4, -1)]
                    v = gp.log_likelihood_restricted(p)
                    raised = False
                    try:
                        gp.log_likelihood_restricted(p, eval_grad=True)
                    except ValueError:
                        raised = True
                    assert raised
                    pars.append(p)
                    vals.append(float(v))
                    n += 1
                key = "T%d_k%d_m%d" % (T, kid, mid)
                out[key + "_par"], out[key + "_llf"] = np.array(pars), np.array(vals)
    assert n == 36
    save("G33_reml_multitarget", **out)


if __name__ == "__main__":
    one("G31_matern_nu08_ok_noisy", 0.8, 3, True, np.r_[0.09, 0.05, 0.12, 0.9], 31)
    one("G32_matern_nu37_sk_noisy", 3.7, 4, False, np.r_[0.05, 0.08, 0.03, 0.06, 0.85], 32)
    golden_reml_multitarget()
