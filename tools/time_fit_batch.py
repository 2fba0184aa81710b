"""A complete GaussianProcess.fit with an evaluation budget of 400 likelihood + gradient evaluations (gpr.py:1058-1197) at N = 512 / 1024 / 2048, d = 20:
the reference-exact sequential loop (scipy L-BFGS-B, one device call per evaluation) against restart_batch = 10 (bogp_mle_batch: the ten restarts
in lock step, one batched device call per round)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bogp

d = 20
for N in (512, 1024, 2048):
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + 3 * np.sin(X[:, 0])
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    for name, kw in (("sequential (scipy, 1 evaluation per device call)", {}), ("restart_batch = 10 (lock step), equal shares", dict(restart_batch=10)),
                     ("restart_batch = 10 (lock step), prune_reserve 20", dict(restart_batch=10, mle_prune_reserve=20))):
        ts, evs, llfs = [], [], []
        for rep in range(3):
            gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-4] * d, thetaU=[1e1] * d, nugget=1e-6,
                                      optimizer="BFGS", wait_iter=10, random_start=10, eval_budget=400, **kw)  # fmt: skip
            np.random.seed(rep)
            t0 = time.perf_counter()
            gp.fit(X, y)
            ts.append(time.perf_counter() - t0)
            evs.append(gp.eval_count)
            llfs.append(gp.log_likelihood_)
        i = int(np.argsort(ts)[1])
        print("N = %4d  %-50s fit %7.1f ms  (%d likelihood evaluations, %.3f ms each%s; llf %.4f)"
              % (N, name, 1e3 * ts[i], evs[i], 1e3 * ts[i] / max(1, evs[i]), ", %d device rounds" % gp.mle_rounds if kw else "", llfs[i]))
