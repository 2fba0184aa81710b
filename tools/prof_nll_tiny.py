"""rocprofv3 target: 200 likelihood + gradient evaluations at N = 32, d = 5 (what the MLE of an early BO iteration costs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bogp import _lib  # noqa: E402

N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 5)
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d))
y = np.sum(X**2, axis=1)
y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
par = np.r_[np.full(d, 0.2 / d), 0.9]
for _ in range(20):
    eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True)
t0 = time.perf_counter()
for _ in range(200):
    eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True)
print("N=%d d=%d: llf + gradient %.1f us per call" % (N, d, (time.perf_counter() - t0) / 200 * 1e6))
eng.close()
