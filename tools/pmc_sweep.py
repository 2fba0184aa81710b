"""One sweep of a bench.py workload and nothing else (no fit timing, no CPU baseline): the process rocprofv3 --pmc is pointed at.
usage: python tools/pmc_sweep.py [C2|C3|C4|C5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from bogp import _lib
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
N, d, M = w["N"], w["d"], w["M"]
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(w["kernel"], _lib.MODE_NOISY, np.r_[np.full(d, w["theta"]), 0.9], 1e-6, False, 0.0)
Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
print(eng.sweep(w["acq"], float(y.min()), True), eng.last_timing())
