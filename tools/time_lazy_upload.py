"""Where the PCIe-inclusive ask() of host-sampled candidates spends its time (C3: N = 2048, d = 20, M = 1e6 = 160 MB of candidates):
resident sweep vs plain upload + sweep vs lazy upload + sweep (bogp_candidates_upload_lazy), pageable and pinned host arrays."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bogp import _lib

N, d, M = 2048, 20, 1_000_000
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d))
y = np.sum(X**2, axis=1)
y = ((y - y.mean()) / y.std()).reshape(-1, 1)
eng = _lib.Engine(0)
eng.set_train(X, y)
eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.01), 0.9], 1e-6, False, 0.0)
acq = [(_lib.ACQ_MGFI, 2.0), (_lib.ACQ_EI, 0.0)]
pl = float(y.min())
Xh = rng.uniform(-5, 5, size=(M, d))
Xp = torch.from_numpy(Xh).pin_memory().numpy()


def med(f, n=5):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


eng.upload_candidates(Xh)
print("sweep, candidates resident:                 %7.2f ms" % med(lambda: eng.sweep(acq, pl, True)))
for name, A in (("pageable", Xh), ("pinned  ", Xp)):
    print("%s plain upload alone:                %7.2f ms" % (name, med(lambda: eng.upload_candidates(A))))
    print("%s plain upload + sweep:              %7.2f ms" % (name, med(lambda: (eng.upload_candidates(A), eng.sweep(acq, pl, True)))))
    t_up = []

    def lazy():
        t0 = time.perf_counter()
        eng.upload_candidates(A, lazy=True)
        t_up.append((time.perf_counter() - t0) * 1e3)
        eng.sweep(acq, pl, True)

    print("%s lazy upload + sweep:               %7.2f ms  (the upload call itself: %.2f ms)" % (name, med(lazy), float(np.median(t_up))))
