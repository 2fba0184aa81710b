"""Latency of the reference-style single-point path: criterion(x, return_dx=True) through the bogp classes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bogp
for (N, d) in [(512, 10), (2048, 20)]:
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    gp = bogp.GaussianProcess(corr="matern", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    gp.set_state(np.r_[np.full(d, 0.02 if N == 512 else 0.01), 0.9], X, y)
    ei = bogp.EI(model=gp)
    xs = rng.uniform(-5, 5, size=(200, d))
    for x in xs[:10]: ei(x.reshape(1, -1), return_dx=True)
    t0 = time.perf_counter()
    for x in xs: ei(x.reshape(1, -1))
    t1 = time.perf_counter()
    for x in xs: ei(x.reshape(1, -1), return_dx=True)
    t2 = time.perf_counter()
    for x in xs: gp.gradient(x)
    t3 = time.perf_counter()
    print("N=%d d=%d: EI(x) %.0f evals/s   EI(x, return_dx) %.0f evals/s   gradient alone %.0f /s" % (N, d, 200 / (t1 - t0), 200 / (t2 - t1), 200 / (t3 - t2)))
