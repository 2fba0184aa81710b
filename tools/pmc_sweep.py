"""One C3 sweep and nothing else (no fit timing, no CPU baseline): the process rocprofv3 --pmc is pointed at."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bogp import _lib
N, d, M = 2048, 20, 1_000_000
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.01), 0.9], 1e-6, False, 0.0)
Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
print(eng.sweep([(_lib.ACQ_MGFI, 2.0), (_lib.ACQ_EI, 0.0)], float(y.min()), True), eng.last_timing())
