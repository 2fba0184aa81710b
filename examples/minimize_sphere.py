"""BASELINE.json configs[0] on the GPU path: minimise sum(x^2) in d = 2 with 30 function evaluations.

A deliberately tiny ask/tell loop (the real drivers are `bayes_optim.BO / ParallelBO`): 5-point
DoE, then per iteration  tell = standardise + GaussianProcess.fit (MLE on the GPU),  ask = EI swept over 50 000
candidates drawn on the device.  The model is what `bayes_optim.fmin` builds (`__init__.py:147-160`): Matern-3/2,
constant trend with estimated beta, nugget 1e-6, BFGS MLE."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bogp


def fmin_sphere(dim=2, max_FEs=30, n_doe=5, seed=42, M=50_000, verbose=False):
    f = lambda x: float(np.sum(np.asarray(x) ** 2))  # noqa: E731
    lo, hi = -5.0, 5.0
    bounds = [(lo, hi)] * dim
    rng = np.random.default_rng(seed)
    np.random.seed(seed)  # the MLE restarts draw from the global stream, like the reference
    X = rng.uniform(lo, hi, size=(n_doe, dim))
    y = np.array([f(x) for x in X])
    rng_len = np.full(dim, hi - lo)
    model = bogp.GaussianProcess(mean=bogp.trend.constant_trend(dim), corr="matern", thetaL=1e-3 * rng_len, thetaU=1e3 * rng_len,
                                 nugget=1e-6, optimizer="BFGS", wait_iter=3, random_start=max(10, dim), eval_budget=100 * dim)  # fmt: skip
    while len(y) < max_FEs:
        ys = (y - y.mean()) / y.std()  # BaseBO.update_model standardises the fitness (base.py:437-441)
        model.fit(X, ys.reshape(-1, 1))
        crit = bogp.EI(model=model, minimize=True, plugin=float(ys.min()))
        _, _, xb = bogp.sweep_generated([crit], bounds, M, seed=int(rng.integers(0, 2**62)))
        x_new = xb[0]
        if np.any(np.all(np.isclose(X, x_new), axis=1)):  # BO.pre_eval_check: never re-evaluate a point
            x_new = rng.uniform(lo, hi, size=dim)
        X = np.vstack([X, x_new])
        y = np.append(y, f(x_new))
        if verbose:
            print("%2d evaluations, best %.6f" % (len(y), y.min()))
    i = int(np.argmin(y))
    return X[i], float(y[i]), len(y)


if __name__ == "__main__":
    xopt, fopt, n = fmin_sphere(verbose=True)
    print("xopt", xopt, "fopt", fopt, "evaluations", n)
