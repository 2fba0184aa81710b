"""Round-2 goldens made by IMPORTING the reference (build container only):  python oracle/make_golden_r02.py

G25_cubic_ok_noisy / G26_genexp_sk_noisy: the two correlation functions of `kernel.py` that the reference can evaluate
but not differentiate (`cubic` :419-466, `generalized_exponential` :332-379; their branches in corr_grad_theta / corr_dx
are `pass`, gpr.py:652-657, 763-766) -- so no `fit`, no gradients, but a pinned state (Appendix A of SURVEY.md), the
posterior, the criteria row by row, np.argmax, and a table of likelihood VALUES in the three estimation modes.
generalized_exponential takes theta = [theta_1 .. theta_d, p] (d + 1 entries); the GP accepts that only with an explicit
trend object of dimension d (its default mean is built from len(thetaU), gpr.py:269-270, and then rejects X at trend.py:57).
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402

from bayes_optim.surrogate import GaussianProcess, trend  # noqa: E402

from oracle.make_golden import acq_rows, make_data, pin, save, state_dict  # noqa: E402

warnings.filterwarnings("ignore")


def llf_values(gp, pars):
    return np.array([float(gp.log_likelihood_concentrated(np.asarray(p, float))) for p in pars])


def one(name, corr, kid, d, n_theta, mean_pin, par, seed, theta_draw):
    X, y = make_data(seed, 70, d)
    y = y + 0.05 * np.random.default_rng(seed + 1).standard_normal(y.shape)
    gp = GaussianProcess(mean=mean_pin(), corr=corr, thetaL=[1e-5] * n_theta, thetaU=[1e2] * n_theta, nugget=1e-6)
    llf = pin(gp, X, y, par)
    rng = np.random.default_rng(seed + 2)
    Xs = rng.uniform(-5, 5, size=(256, d))
    Xs[5] = X[9]  # a candidate on a training point
    mu, mse = gp.predict(Xs, eval_MSE=True)
    tabs = {}
    rng2 = np.random.default_rng(seed + 3)
    for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
        for tname, ok in (("sk", False), ("ok", True)):
            g2 = GaussianProcess(mean=trend.constant_trend(d) if ok else trend.constant_trend(d, beta=0), corr=corr,
                                 thetaL=[1e-5] * n_theta, thetaU=[1e2] * n_theta, **kw)  # fmt: skip
            g2._check_data(X, y)
            pars = []
            for _ in range(4):
                th = theta_draw(rng2)
                pars.append(th if mid == 0 else np.r_[th, rng2.uniform(0.4, 1.1) if mid == 1 else rng2.uniform(0.7, 0.999)])
            key = "t_m%d_%s" % (mid, tname)
            tabs[key + "_par"], tabs[key + "_llf"] = np.array(pars), llf_values(g2, pars)
    save(name, par=np.asarray(par, float), Xs=Xs, mu=mu, mse=mse, kernel=np.array(kid), mode=np.array(1),
         **state_dict(gp, llf), **acq_rows(gp, Xs), **tabs)  # fmt: skip


def golden_reml_trends():
    """G27: the restricted likelihood (gpr.py:813-918) with the linear (p = d + 1) and quadratic (p = (d+1)(d+2)/2) trend
    bases, value + gradient, three modes x {estimated, fixed coefficients} x {SE, Matern-3/2}: the p > 1 forms of
    -log det(F^T F), log prod diag(G)^2 and the (L^-T Q)(L^-T Q)^T gradient term."""
    X, y = make_data(27, 45, 3)
    y = y + 0.2 * np.random.default_rng(327).standard_normal(y.shape)
    d = 3
    out = dict(X=X, y=y)
    rng = np.random.default_rng(227)
    n = 0
    for tid, tcls in ((1, trend.linear_trend), (2, trend.quadratic_trend)):
        p = d + 1 if tid == 1 else (d + 1) * (d + 2) // 2
        beta_fixed = np.round(rng.uniform(-0.3, 0.3, size=p), 3)
        out["t%d_beta" % tid] = beta_fixed
        for kid, corr in ((0, "squared_exponential"), (2, "matern")):
            for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
                for tname, mean in (("uk", lambda: tcls(d)), ("sk", lambda: tcls(d, beta=beta_fixed))):
                    gp = GaussianProcess(mean=mean(), corr=corr, thetaL=[1e-4] * d, thetaU=[1e2] * d, likelihood="restricted", **kw)
                    gp._check_data(X, y)
                    pars, vals, grads = [], [], []
                    for _ in range(3):
                        th = 10 ** rng.uniform(-1.5, -0.6, size=d)
                        pr = np.r_[th, rng.uniform(0.3, 1.2)]
                        if mid == 2:
                            pr = np.r_[pr, 10 ** rng.uniform(-4, -1)]
                        v, g = gp.log_likelihood_restricted(pr, eval_grad=True)
                        pars.append(pr)
                        vals.append(float(v))
                        grads.append(np.asarray(g, float).ravel())
                        n += 1
                    key = "t%d_k%d_m%d_%s" % (tid, kid, mid, tname)
                    out[key + "_par"], out[key + "_llf"], out[key + "_grad"] = np.array(pars), np.array(vals), np.array(grads)
    assert n == 72
    save("G27_reml_trend_tables", **out)


if __name__ == "__main__":
    golden_reml_trends()
    d = 4
    one("G25_cubic_ok_noisy", "cubic", 5, d, d, lambda: trend.constant_trend(d), np.r_[0.06, 0.09, 0.05, 0.08, 0.85], 25,
        lambda r: 10 ** r.uniform(-1.5, -0.8, size=d))
    d = 3
    one("G26_genexp_sk_noisy", "generalized_exponential", 6, d, d + 1, lambda: trend.constant_trend(d, beta=0),
        np.r_[0.11, 0.07, 0.16, 1.6, 0.9], 26, lambda r: np.r_[10 ** r.uniform(-1.3, -0.5, size=d), r.uniform(1.0, 2.0)])
