import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from bogp import _lib
eng = _lib.Engine(0)
N, d = 8192, 50
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
par = np.r_[np.full(d, 0.004), 0.9]
eng.set_train(X, y)
for _ in range(3):
    eng.nll(0, 1, par, 1e-6, False, 0.0, eval_grad=True)
