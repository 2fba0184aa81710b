"""Round-5 golden made by IMPORTING the reference (build container only):  python oracle/make_golden_r05.py

G37_matern_nu_illcond: the general-nu Matern arm (kernel.py:201-207, scipy.special.kv) on an ILL-CONDITIONED correlation matrix
(cond(R) ~ 1e9 .. 1e10, noiseless mode, ordinary kriging) -- VERDICT r04 "weak" item 1.  Beside the reference's outputs (pinned state,
posterior at 96 candidates) the fixture holds the TRUE posterior of the very same model: R and r built from mpmath's K_nu at 50 digits,
every solve in 50-digit arithmetic, the reference's formulas (gpr.py:486-510, 790-811, 920-992) in exact arithmetic.  scipy's kv is up
to hundreds of eps from the true K_nu (tests/golden/G36_kv_table.npz), so the reference's R is a perturbed matrix and cond(R) amplifies
the perturbation: the reference is `ref_err_*` away from its own model's exact answer.  The GPU test holds the device to the TRUTH at the
generic tolerance of an ill-conditioned problem (100 cond eps) and to the reference at that plus the reference's own distance from the
truth.  A well-conditioned twin (same data, larger theta) certifies the exact-arithmetic formulas against the reference to 1e-11."""
import functools
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import mpmath as mp  # noqa: E402
import numpy as np  # noqa: E402

from bayes_optim.surrogate import GaussianProcess, trend  # noqa: E402
from bayes_optim.surrogate.gaussian_process.kernel import matern  # noqa: E402

from oracle.make_golden import make_data, pin, save, state_dict  # noqa: E402

warnings.filterwarnings("ignore")
mp.mp.dps = 50


def true_profile(nu, s2):
    if s2 == 0:
        return mp.mpf(1)
    t = mp.sqrt(2 * nu) * mp.sqrt(s2)
    return mp.mpf(2) ** (1 - nu) / mp.gamma(nu) * t**nu * mp.besselk(nu, t)


def exact_posterior(X, y, theta, nu, Xs):
    """Ordinary kriging, noiseless mode, in 50-digit arithmetic: beta = 1'R^-1 y / 1'R^-1 1, sigma2 = (y - beta)' R^-1 (y - beta) / (N - 1)
    (gpr.py:941-951 with k = rank(Q Q') = 1), mu = beta + r' R^-1 (y - beta), MSE = sigma2 (1 - r' R^-1 r + (1' R^-1 r - 1)^2 / 1' R^-1 1)."""
    N, d = X.shape
    nu = mp.mpf(float(nu))
    th = [mp.mpf(float(t)) for t in theta]
    Xm = [[mp.mpf(float(v)) for v in row] for row in X]
    R = mp.matrix(N, N)
    for i in range(N):
        R[i, i] = 1
        for j in range(i):
            s2 = sum(th[k] * (Xm[i][k] - Xm[j][k]) ** 2 for k in range(d))
            R[i, j] = R[j, i] = true_profile(nu, s2)
    ym = mp.matrix([mp.mpf(float(v)) for v in y.ravel()])
    one = mp.matrix([1] * N)
    Riy, Ri1 = mp.lu_solve(R, ym), mp.lu_solve(R, one)
    s11 = (one.T * Ri1)[0]
    beta = (one.T * Riy)[0] / s11
    res = ym - beta * one
    Rires = mp.lu_solve(R, res)
    sigma2 = (res.T * Rires)[0] / (N - 1)
    mu, mse = [], []
    for xs in Xs:
        xm = [mp.mpf(float(v)) for v in xs]
        r = mp.matrix([true_profile(nu, sum(th[k] * (xm[k] - Xm[i][k]) ** 2 for k in range(d))) for i in range(N)])
        Rir = mp.lu_solve(R, r)
        u = (one.T * Rir)[0] - 1
        mu.append(float(beta + (r.T * Rires)[0]))
        mse.append(float(sigma2 * (1 - (r.T * Rir)[0] + u * u / s11)))
    # 2-norm condition number of R from its double rounding (what both implementations factorise)
    w = np.linalg.eigvalsh(np.array(R.tolist(), dtype=float))
    return np.array(mu), np.array(mse), float(beta), float(sigma2), float(w[-1] / w[0])


def make(nu, d, N, theta, seed):
    corr = functools.partial(matern, nu=nu)
    X, y = make_data(seed, N, d)
    y = y + 0.02 * np.random.default_rng(seed + 1).standard_normal(y.shape)
    gp = GaussianProcess(mean=trend.constant_trend(d), corr=corr, thetaL=[1e-6] * d, thetaU=[1e2] * d, nugget=0)
    llf = pin(gp, X, y, np.asarray(theta, float))
    Xs = np.random.default_rng(seed + 2).uniform(-5, 5, size=(96, d))
    mu, mse = gp.predict(Xs, eval_MSE=True)
    tmu, tmse, tbeta, ts2, cond = exact_posterior(X, y, theta, nu, Xs)
    return gp, llf, Xs, mu[:, 0], mse[:, 0], tmu, tmse, tbeta, ts2, cond


if __name__ == "__main__":
    nu, d, N, seed = 3.7, 2, 60, 37
    # well-conditioned twin: the exact-arithmetic formulas ARE the reference's
    gp, llf, Xs, mu, mse, tmu, tmse, tbeta, ts2, cond = make(nu, d, N, [0.9, 1.3], seed)
    print("twin: cond %.2e, |mu - true| %.2e, |mse - true| / sigma2 %.2e, beta %.2e, sigma2 %.2e" % (
        cond, np.abs(mu - tmu).max(), np.abs(mse - tmse).max() / ts2, abs(float(np.ravel(gp.mean.beta)[0]) - tbeta), abs(float(gp.sigma2[0]) - ts2) / ts2))
    assert cond < 1e5 and np.abs(mu - tmu).max() < 1e-10 and np.abs(mse - tmse).max() < 1e-10 * ts2
    theta = [0.018, 0.027]  # cond(R) = 7.9e9
    gp, llf, Xs, mu, mse, tmu, tmse, tbeta, ts2, cond = make(nu, d, N, theta, seed)
    ref_err_mu = float(np.abs(mu - tmu).max())
    ref_err_mse = float(np.abs(mse - tmse).max() / ts2)
    print("ill-conditioned: cond %.2e, reference vs truth: |mu| %.2e, |mse| / sigma2 %.2e, sigma2 rel %.2e" % (
        cond, ref_err_mu, ref_err_mse, abs(float(gp.sigma2[0]) - ts2) / ts2))
    assert cond >= 1e9
    par_engine = np.r_[theta, nu]  # noiseless mode: [theta_1 .. theta_d, nu]
    save("G37_matern_nu_illcond", par=par_engine, nu=np.array(nu), Xs=Xs, mu=mu, mse=mse, kernel=np.array(7), mode=np.array(0),
         true_mu=tmu, true_mse=tmse, true_beta=np.array(tbeta), true_sigma2=np.array(ts2), cond=np.array(cond),
         ref_err_mu=np.array(ref_err_mu), ref_err_mse=np.array(ref_err_mse), **state_dict(gp, llf))


def golden_reml_isotropic():
    """G38: the restricted likelihood's GRADIENT with an isotropic theta (thetaL of length 1 on d = 1 / 2 / 3 / 4 inputs), three modes x
    {simple, ordinary kriging} x {SE, Matern-3/2}.  The reference builds the (N, N, d) tensor of per-dimension derivatives whatever
    len(theta) is (gpr.py:736-770), appends R0 [and I] and reads slice i for parameter i (:889-900): for d >= 2 the entries of the returned
    vector are the first n_par per-dimension slices, not (d/dtheta, d/dsigma2[, d/dnoise]).  That is what the golden holds and what the
    device returns (VERDICT r04 "missing" item 4: it refused the call)."""
    rng = np.random.default_rng(538)
    out, n = {}, 0
    for d in (1, 2, 3, 4):
        X, y = make_data(38 + d, 36, d)
        y = y + 0.2 * rng.standard_normal(y.shape)
        out["X%d" % d], out["y%d" % d] = X, y
        for kid, corr in ((0, "squared_exponential"), (2, "matern")):
            for mid, kw in ((0, dict(nugget=0)), (1, dict(nugget=1e-6)), (2, dict(nugget=1e-6, noise_estim=True))):
                for tname in ("sk", "ok"):  # (mean=None would size the default trend by len(thetaL) = 1: gpr.py:269-270)
                    mean = trend.constant_trend(d) if tname == "ok" else trend.constant_trend(d, beta=0)
                    gp = GaussianProcess(mean=mean, corr=corr, thetaL=[1e-4], thetaU=[1e2], likelihood="restricted", **kw)
                    gp._check_data(X, y)
                    pars, vals, grads = [], [], []
                    while len(pars) < 3:
                        p = np.r_[10 ** rng.uniform(-1.2, -0.3) * {1: 200.0, 2: 4.0, 3: 1.5, 4: 1.0}[d], rng.uniform(0.3, 1.2)]  # (36 points: well conditioned in every d)
                        if mid == 2:
                            p = np.r_[p, 10 ** rng.uniform(-4, -1)]
                        v, g = gp.log_likelihood_restricted(p, eval_grad=True)
                        if not np.isfinite(v):  # a positive value is rejected (-inf, :868-871): draw again
                            continue
                        assert len(np.ravel(g)) == len(p), (d, kid, mid, tname, np.shape(g))
                        pars.append(p), vals.append(float(v)), grads.append(np.asarray(g, float).ravel())
                        n += 1
                    key = "d%d_k%d_m%d_%s" % (d, kid, mid, tname)
                    out[key + "_par"], out[key + "_llf"], out[key + "_grad"] = np.array(pars), np.array(vals), np.array(grads)
    assert n == 4 * 2 * 3 * 2 * 3
    save("G38_reml_isotropic", **out)
