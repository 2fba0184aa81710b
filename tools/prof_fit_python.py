import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bogp
N, d = 64, 10
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = (y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)
gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-3] * d, thetaU=[1e2] * d, nugget=1e-6, random_start=10)
np.random.seed(1); gp.fit(X, y.reshape(-1, 1))
np.random.seed(1)
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); gp.fit(X, y.reshape(-1, 1)); t1 = time.perf_counter(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print("fit wall %.1f ms" % ((t1 - t0) * 1e3)); print(s.getvalue()[:3500])
