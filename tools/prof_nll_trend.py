"""rocprofv3 target: 5 likelihood + gradient evaluations with the linear (argv[1] = 1) or quadratic (2) trend basis at N = 2048, d = 20."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
tid = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, d = 2048, 20
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1) + 0.5 * X[:, 0]; y = ((y - y.mean()) / y.std()).reshape(-1, 1)
par = np.r_[np.full(d, 0.01), 0.9]
eng = _lib.Engine(0)
eng.set_train(X, y)
for _ in range(5):
    out = eng.nll(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=tid)
print(out[0])
