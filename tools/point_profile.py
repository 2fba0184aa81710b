"""rocprofv3 target: 300 one-point calls + 20 batched calls (B = 32) at C3 size; see profiles/r03_point_*_kernel_stats.csv."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bogp import _lib  # noqa: E402

N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 20)
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d))
y = np.sum(X**2, axis=1)
y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
acq, pl = [(_lib.ACQ_EI, 0.0)], float(y.min())
x = rng.uniform(-5, 5, size=d)
for _ in range(300):
    eng.point_eval(x, acq, pl, True)
Xb = rng.uniform(-5, 5, size=(32, d))
for _ in range(20):
    eng.point_eval_batch(Xb, acq, pl, True)
eng.close()
