"""Where the time of the fused small-N sweep (k_sweep_small) goes at the C2 size: producer-only launches (predict without
MSE), full launches, and the dependence on M (workgroup rounds: 256 CUs x one 64-candidate workgroup each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bogp import _lib

N, d = int(os.environ.get("N", 512)), int(os.environ.get("D", 10))
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(_lib.KERNEL_SE, _lib.MODE_NOISY, np.r_[np.full(d, 0.02), 0.9], 1e-6, False, 0.0)
flops = float(N) * N + 3.0 * N
for M in (16384, 32768, 65536, 98304, 100000, 114688, 131072, 1048576):
    Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
    eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
    out = []
    for tag, fn in (("mean-only", lambda: eng.predict(eval_MSE=False)), ("sweep", lambda: eng.sweep([(_lib.ACQ_EI, 0.0)], float(y.min()), True))):
        for _ in range(3): fn()
        ts = []
        for _ in range(10):
            fn(); ts.append(eng.last_timing()["contract_ms"] + eng.last_timing()["corr_ms"] + eng.last_timing()["acquisition_ms"])
        out.append((tag, float(np.median(ts))))
    rounds = (M + 63) // 64 / 256.0
    print("M %8d  rounds %6.2f  mean-only %.3f ms  sweep %.3f ms  -> %.1f TF/s on the contraction flops" % (M, rounds, out[0][1], out[1][1], flops * M / out[1][1] / 1e9))
