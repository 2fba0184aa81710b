"""rocprofv3 target: 3 likelihood + gradient evaluations at N (argv[1], default 8192), d = 50."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
N, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 8192), 50
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
par = np.r_[np.full(d, 0.004), 0.9]
eng.set_train(X, y)
for _ in range(3):
    out = eng.nll(0, 1, par, 1e-6, False, 0.0, eval_grad=True)
print("N=%d llf %.6f" % (N, out[0]))
