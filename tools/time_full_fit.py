"""Wall time of a complete GaussianProcess.fit (the reference's L-BFGS-B restart loop on the host, every likelihood +
gradient on the device) at BASELINE.json's model sizes.  The reference needs 10.9-17 s PER evaluation at N = 2048
(SURVEY section 6), i.e. hours per fit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bogp

bogp._lib.Engine(0).close()  # library / rocBLAS initialisation is not part of a fit
warm = bogp.GaussianProcess(thetaL=[1e-2] * 2, thetaU=[1e1] * 2, nugget=1e-6, eval_budget=10)
warm.fit(np.random.default_rng(9).uniform(-1, 1, (70, 2)), np.random.default_rng(9).standard_normal((70, 1)))
for (N, d, corr, budget) in [(512, 10, "squared_exponential", None), (2048, 20, "matern52", None), (2048, 20, "matern52", 400)]:
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1) + 5.0 * rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr=corr, thetaL=[1e-4] * d, thetaU=[1e1] * d, nugget=1e-6,
                              random_start=3, wait_iter=3, eval_budget=budget)
    np.random.seed(1)
    t0 = time.perf_counter(); gp.fit(X, y); t = time.perf_counter() - t0
    print("N=%5d d=%2d %-20s budget %-5s: fit %.2f s, %d likelihood+gradient evaluations (%.2f ms each incl. host), llf %.3f"
          % (N, d, corr, budget if budget else "200*n_par", t, gp.eval_count, 1e3 * t / max(1, gp.eval_count), gp.log_likelihood_))
