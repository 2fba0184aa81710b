"""G24: an ENSEMBLE of complete `GaussianProcess.fit` runs by the imported reference.

Run (build container only):   python oracle/make_fit_golden.py      -> tests/golden/G24_fit_ensemble.npz

Why an ensemble: the reference's MLE hands L-BFGS-B the gradient d llf / d par for an objective of log10(par)
(`gpr.py:1113-1123`), so its line search is not a descent on a consistent function and amplifies last-bit differences
of the likelihood into different restart outcomes.  One fit therefore proves nothing about another implementation's
`fit`; a distribution does.  For 54 seeded problems (N in {20, 50, 100} x d in {2, 3, 5} x the three estimation modes
x two seeds, SE / Matern-3/2, simple / ordinary kriging alternating) this file holds

  * the inputs (X, y, the constructor keywords, the np.random seed the fit starts from),
  * the reference's fitted theta / sigma2 / noise_var, its final log-likelihood and its likelihood-evaluation count,
  * the NULL distribution: the same host loop run three more times on the oracle with the likelihood value and gradient
    perturbed by a relative 1e-13 that is a deterministic pseudo-random function of the parameter vector (what ANY
    implementation with another summation order looks like to the optimiser): `null_llf[case, 3]`.

`tests/test_gpu_parity.py::test_fit_ensemble_matches_the_reference_distribution` fits the same problems on the device
and asserts that the paired differences of the final log-likelihood are centred and no more one-sided than the null.
"""
import hashlib
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))

import numpy as np  # noqa: E402
import scipy  # noqa: E402

import bayes_optim  # noqa: E402,F401
from bayes_optim.surrogate import GaussianProcess, trend  # noqa: E402

import bogp  # noqa: E402
from support.oracle_engine import OracleEngine  # noqa: E402

warnings.filterwarnings("ignore")
OUT = os.path.join(ROOT, "tests", "golden", "G24_fit_ensemble.npz")
MODES = ("noiseless", "noisy", "noise_estim")


def cases():
    i = 0
    for N in (20, 50, 100):
        for d in (2, 3, 5):
            for mode in MODES:
                for rep in (0, 1):
                    yield dict(case=i, N=N, d=d, mode=mode, corr=("squared_exponential", "matern")[(i // 2) % 2],
                               ok=bool((i // 3) % 2), data_seed=1000 + i, fit_seed=7 + i)  # fmt: skip
                    i += 1


def problem(c):
    rng = np.random.default_rng(c["data_seed"])
    X = rng.uniform(-5, 5, size=(c["N"], c["d"]))
    y = np.sum(X**2, axis=1) + np.sin(3.0 * X[:, 0])
    y = (y - y.mean()) / y.std() + 0.1 * rng.standard_normal(c["N"])
    return X, y.reshape(-1, 1)


def ctor_kwargs(c):
    d = c["d"]
    return dict(corr=c["corr"], thetaL=[1e-3] * d, thetaU=[1e2] * d, nugget=0 if c["mode"] == "noiseless" else 1e-6,
                noise_estim=c["mode"] == "noise_estim", optimizer="BFGS", wait_iter=3, random_start=5, eval_budget=100 * d)  # fmt: skip


class PerturbedOracleEngine(OracleEngine):
    """The oracle with its likelihood value and gradient multiplied by (1 + 1e-13 xi), xi ~ N(0, 1) drawn from a generator
    seeded by (variant, the bytes of par): a DETERMINISTIC function of the parameters, like the rounding error of any
    implementation (a repeated evaluation returns the same bits), but unrelated to the reference's own rounding --
    the null model of 'a correct implementation with a different summation order'."""

    def __init__(self, variant):
        super().__init__()
        self._variant = int(variant)

    def nll(self, kernel, mode, par, *a, **kw):
        out = super().nll(kernel, mode, par, *a, **kw)
        key = int.from_bytes(hashlib.blake2b(np.ascontiguousarray(par, dtype=np.float64).tobytes(), digest_size=8).digest(), "little")
        rng = np.random.default_rng([self._variant, key])
        if isinstance(out, tuple):
            return out[0] * (1 + 1e-13 * rng.standard_normal()), out[1] * (1 + 1e-13 * rng.standard_normal(out[1].shape))
        return out * (1 + 1e-13 * rng.standard_normal())


def main():
    out, n = {}, 0
    ref_llf, null_llf, evals = [], [], []
    for c in cases():
        X, y = problem(c)
        kw = ctor_kwargs(c)
        gp = GaussianProcess(mean=trend.constant_trend(c["d"]) if c["ok"] else None, **kw)
        count = [0]
        orig = gp.log_likelihood_concentrated

        def counted(par, env=None, eval_grad=False, _o=orig, _c=count):
            _c[0] += 1
            return _o(par, env, eval_grad)

        gp.log_likelihood_concentrated = counted
        np.random.seed(c["fit_seed"])
        gp.fit(X, y)
        nl = []
        for s in range(3):
            g2 = bogp.GaussianProcess(mean=bogp.trend.constant_trend(c["d"]) if c["ok"] else None, **kw)
            g2._engine = PerturbedOracleEngine(100 * c["case"] + s)
            np.random.seed(c["fit_seed"])
            g2.fit(X, y)
            nl.append(float(g2.log_likelihood_))
        # the un-perturbed oracle replays the reference exactly (same optimum, bit for bit)
        g3 = bogp.GaussianProcess(mean=bogp.trend.constant_trend(c["d"]) if c["ok"] else None, **kw)
        g3._engine = OracleEngine()
        np.random.seed(c["fit_seed"])
        g3.fit(X, y)
        assert g3.estimation_mode == gp.estimation_mode
        np.testing.assert_allclose(g3.log_likelihood_, gp.log_likelihood_, rtol=1e-9)
        k = "c%02d_" % c["case"]
        out.update({k + "X": X, k + "y": y, k + "theta": gp.theta_, k + "sigma2": np.atleast_1d(gp.sigma2).astype(float),
                    k + "noise_var": np.atleast_1d(gp.noise_var).astype(float)})  # fmt: skip
        ref_llf.append(float(gp.log_likelihood_))
        null_llf.append(nl)
        evals.append(count[0])
        out[k + "final_mode"] = np.array(MODES.index(gp.estimation_mode))
        print(c, "llf %.6f evals %d null %s" % (gp.log_likelihood_, count[0], np.round(np.array(nl) - gp.log_likelihood_, 6)), flush=True)
        n += 1
    cs = list(cases())
    out.update(n_cases=np.array(n), N=np.array([c["N"] for c in cs]), d=np.array([c["d"] for c in cs]),
               mode=np.array([MODES.index(c["mode"]) for c in cs]), corr=np.array([c["corr"] == "matern" for c in cs]),
               ok=np.array([c["ok"] for c in cs]), fit_seed=np.array([c["fit_seed"] for c in cs]),
               ref_llf=np.array(ref_llf), null_llf=np.array(null_llf), ref_evals=np.array(evals),
               ver_numpy=np.array(np.__version__), ver_scipy=np.array(scipy.__version__))  # fmt: skip
    np.savez_compressed(OUT, **out)
    dn = np.array(null_llf) - np.array(ref_llf)[:, None]
    print("null: |d| > 1e-6 in %d of %d perturbed fits; median %.3g; perturbed better in %d, worse in %d"
          % (np.sum(np.abs(dn) > 1e-6), dn.size, np.median(dn), np.sum(dn > 1e-6), np.sum(dn < -1e-6)))


if __name__ == "__main__":
    main()
