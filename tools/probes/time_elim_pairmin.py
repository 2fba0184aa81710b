import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
for N in [int(a) for a in sys.argv[1:]] or (1152, 1216, 1280, 1344, 1408, 1472):
    d = 20
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=False)
    t0 = time.perf_counter()
    for _ in range(30): r = eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=False)
    t = (time.perf_counter() - t0) / 30 * 1e6
    nb = (N + 63) // 64
    print("N=%d (nb=%d, %d blocks a step): llf %.0f us = %.1f us a step" % (N, nb, (nb + 1) * (nb + 2) // 2 - 1, t, t / nb))
