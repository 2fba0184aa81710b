import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bogp import _lib
N, d = 512, 10
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0); eng.set_train(X, y)
eng.commit(_lib.KERNEL_SE, _lib.MODE_NOISY, np.r_[np.full(d, 0.02), 0.9], 1e-6, False, 0.0)
for mr in (2, 3, 4):
    for rounds in (1, 4):
        M = 256 * 16 * mr * rounds
        Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
        eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
        if mr < 4: os.environ["BOGP_SMALL_FORCE_MR"] = str(mr)
        ts = []
        for i in range(8):
            eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True); ts.append(eng.last_timing()["contract_ms"])
        os.environ.pop("BOGP_SMALL_FORCE_MR", None)
        print("MR %d rounds %d: %.1f us per launch" % (mr, rounds, 1e3 * float(np.median(ts[2:]))))
