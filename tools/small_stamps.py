import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bogp import _lib
N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 10)
print("== N = %d, d = %d (SE, EI)" % (N, d), flush=True)
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0); eng.set_train(X, y)
eng.commit(_lib.KERNEL_SE, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, False, 0.0)
for M in (16384, 98304):
    Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
    eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
    for _ in range(3): eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True)
    os.environ["BOGP_SMALL_STAMPS"] = "1"
    eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True)
    eng.predict(eval_MSE=False)
    del os.environ["BOGP_SMALL_STAMPS"]
