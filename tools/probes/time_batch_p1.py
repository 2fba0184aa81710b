"""One evaluation through bogp_nll_batch's launch plans (P = 1 .. 3) against the sequential bogp_nll: is the grouped chain of the batch
(BOGP_ELIM_SPLIT_BLOCKS=1 forces it at P = 1) shorter than the fused one-launch-a-step chain?  The plan is read once per process: run
this script once per environment."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib

eng = _lib.Engine(0)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
for N, d in ((512, 10), (1024, 20), (2048, 20)):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
    t0 = time.perf_counter()
    for _ in range(30): r = eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
    seq = (time.perf_counter() - t0) / 30 * 1e6
    out = []
    for P in (1, 2, 3, 4):
        pars = np.vstack([par * (1 + 0.1 * s) for s in range(P)])
        eng.nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        t0 = time.perf_counter()
        for _ in range(30): bl, bg, bi = eng.nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        out.append((time.perf_counter() - t0) / 30 * 1e6)
        assert bi[0] == 0 and bl[0] == r[0], (bl[0], r[0])
    print("N=%d: sequential %.0f us; batch P=1..4: %s us" % (N, seq, " ".join("%.0f" % o for o in out)))
