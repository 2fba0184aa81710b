"""r06: where a default `tell` (GaussianProcess.fit: the reference's sequential multi-restart L-BFGS-B loop, scipy, look-ahead by size) spends its wall time at the
sizes of an ordinary BO run -- calls and seconds inside Engine.nll (ctypes call + device) against everything around it (scipy's driver, the objective wrapper).
usage: python tools/ab/r06_tell_breakdown.py [lookahead]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bogp
from bogp import _lib

look = int(sys.argv[1]) if len(sys.argv) > 1 else -1
acc = {"n": 0, "t": 0.0}
lock = threading.Lock()
orig = _lib.Engine.nll
def timed(self, *a, **k):
    t0 = time.perf_counter()
    try:
        return orig(self, *a, **k)
    finally:
        dt = time.perf_counter() - t0
        with lock:
            acc["n"] += 1; acc["t"] += dt
_lib.Engine.nll = timed
d = 10
f = lambda x: float(10 * len(x) + np.sum(np.asarray(x) ** 2 - 10 * np.cos(2 * np.pi * np.asarray(x))))
rng = np.random.default_rng(1); np.random.seed(1)
for N in (50, 100, 200, 400):
    X = rng.uniform(-5.12, 5.12, size=(N, d)); y = np.array([f(x) for x in X]); ys = ((y - y.mean()) / y.std()).reshape(-1, 1)
    rl = np.full(d, 10.24)
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=1e-3 * rl, thetaU=1e3 * rl, nugget=1e-6, optimizer="BFGS",
                              wait_iter=3, random_start=max(10, d), eval_budget=100 * d, restart_lookahead=look)
    gp.fit(X, ys)  # warm-up (allocations)
    acc["n"], acc["t"] = 0, 0.0
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gp.fit(X, ys)
    wall = (time.perf_counter() - t0) / reps
    n, t = acc["n"] / reps, acc["t"] / reps
    print("N = %4d lookahead %2d: fit %.1f ms; %.0f likelihood evaluations, %.1f ms inside Engine.nll (%.0f us each, threads overlap), %.1f ms elsewhere; device-only floor ~ %.1f ms"
          % (N, look, wall * 1e3, n, t * 1e3, t / max(n, 1) * 1e6, (wall - t) * 1e3 if look == 0 else float("nan"), t * 1e3))
