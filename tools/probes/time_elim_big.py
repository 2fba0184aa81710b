"""Where the 64-block elimination stops paying against the Cholesky + inverse path above N = 2048 (BOGP_NLL_ELIM_MAX, read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
for N in (2112, 2304, 2560, 3072, 3584, 4096):
    d = 20
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    eng.set_train(X, y)
    out = []
    for grad in (False, True):
        eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=grad)
        t0 = time.perf_counter()
        for _ in range(10): r = eng.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=grad)
        out.append((time.perf_counter() - t0) / 10 * 1e6)
    print("N=%d: llf %.0f us, llf+grad %.0f us (llf %.6f)" % (N, out[0], out[1], r[0]))
