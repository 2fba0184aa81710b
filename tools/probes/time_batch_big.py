"""bogp_nll_batch above N = 2048: the helper-handle path (default) against the batched elimination (BOGP_NLL_ELIM_MAX raised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
for N in (2304, 2560, 3072):
    d = 20
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    for P in (2, 4, 10):
        pars = np.vstack([par * (1 + 0.05 * s) for s in range(P)])
        eng.nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        t0 = time.perf_counter()
        for _ in range(5): bl, bg, bi = eng.nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        t = (time.perf_counter() - t0) / 5 * 1e6
        print("N=%d P=%d: %.0f us a batch = %.0f us an evaluation (llf[0] %.6f)" % (N, P, t, t / P, bl[0]))
