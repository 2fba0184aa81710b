"""Likelihood(+gradient) latency at the training-set sizes of an ordinary BO run (N = 16 ... 1024), where the evaluation
is launch-latency bound; and a complete fit (multi-restart L-BFGS-B, gpr.py:1058-1197) at each size."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bogp  # noqa: E402
from bogp import _lib  # noqa: E402


def main():
    eng = _lib.Engine(0)
    for N, d in ((16, 2), (32, 5), (64, 5), (100, 5), (128, 5), (144, 10), (156, 10), (160, 10), (200, 10), (252, 10), (256, 10), (512, 10), (1024, 20)):
        rng = np.random.default_rng(0)
        X = rng.uniform(-5, 5, size=(N, d))
        y = np.sum(X**2, axis=1)
        y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
        par = np.r_[np.full(d, 0.2 / d), 0.9]
        eng.set_train(X, y)
        eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
        out = []
        for grad in (False, True):
            n = 200
            t0 = time.perf_counter()
            for _ in range(n):
                eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=grad)
            out.append((time.perf_counter() - t0) / n * 1e6)
        gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-3] * d, thetaU=[1e2] * d, nugget=1e-6)
        np.random.seed(1)
        t0 = time.perf_counter()
        gp.fit(X, y)
        t_fit = time.perf_counter() - t0
        print("N=%4d d=%2d: llf %.0f us, llf+grad %.0f us, full fit %.3f s" % (N, d, out[0], out[1], t_fit))


if __name__ == "__main__":
    main()
