"""Seconds per likelihood(+gradient) evaluation and per commit at pinned parameters (fit is reported separately)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
for (N, d, kernel, th) in [(512, 10, 0, 0.02), (2048, 20, 2, 0.01), (2048, 20, 3, 0.01), (8192, 50, 0, 0.004)]:
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, th), 0.9]
    t0 = time.perf_counter(); eng.set_train(X, y); t_set = time.perf_counter() - t0
    eng.nll(kernel, 1, par, 1e-6, False, 0.0, eval_grad=True)  # warm-up (library init)
    t0 = time.perf_counter(); llf = eng.nll(kernel, 1, par, 1e-6); t_llf = time.perf_counter() - t0
    t0 = time.perf_counter(); llf, g = eng.nll(kernel, 1, par, 1e-6, False, 0.0, eval_grad=True); t_grad = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.commit(kernel, 1, par, 1e-6, False, 0.0); t_commit = time.perf_counter() - t0
    print("N=%5d d=%2d kernel=%d: set_train %.3fs  llf %.4fs  llf+grad %.4fs  commit %.4fs  (llf=%.3f)" % (N, d, kernel, t_set, t_llf, t_grad, t_commit, llf))
