"""A complete lock-step fit (restart_batch = 10, 400-evaluation budget, as tools/time_fit_batch.py) above N = 2048: the batched elimination (default since the
end of r05, N <= 3072) against the helper-handle path (run with BOGP_NLL_ELIM_MAX=2048), and the sequential default loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bogp
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
d = 20
for N in (2560, 3072):
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + 3 * np.sin(X[:, 0])
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    for name, kw in (("sequential (scipy)", {}), ("restart_batch = 10 (lock step)", dict(restart_batch=10))):
        ts = []
        for rep in range(2):
            gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-4] * d, thetaU=[1e1] * d, nugget=1e-6,
                                      optimizer="BFGS", wait_iter=10, random_start=10, eval_budget=400, **kw)  # fmt: skip
            np.random.seed(rep)
            t0 = time.perf_counter()
            gp.fit(X, y)
            ts.append(time.perf_counter() - t0)
        print("N = %4d  %-32s fit %7.1f ms  (%d likelihood evaluations; llf %.4f)" % (N, name, 1e3 * min(ts), gp.eval_count, gp.log_likelihood_))
