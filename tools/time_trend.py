"""Sweep throughput and fit cost with polynomial trends (p > 1) at the C3 size: the universal-kriging term adds two library
GEMMs per chunk (T = r W: 2 N p flops per candidate; C S: 2 p^2) beside k_contract's N^2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bogp import _lib

N, d, M = 2048, 20, 1_000_000
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1) + 0.5 * X[:, 0]; y = ((y - y.mean()) / y.std()).reshape(-1, 1)
y = y + 0.05 * rng.standard_normal(y.shape)
par = np.r_[np.full(d, 0.01), 0.9]
eng = _lib.Engine(0)
eng.set_train(X, y)
Xs = (torch.rand((M, d), dtype=torch.float64, device="cuda") * 10 - 5).contiguous()
eng.bind_candidates(Xs.data_ptr(), M, owner=Xs)
acq = [(_lib.ACQ_MGFI, 2.0), (_lib.ACQ_EI, 0.0)]
for name, tid in (("constant (p=1)", 0), ("linear (p=21)", 1), ("quadratic (p=231)", 2)):
    eng.nll(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=tid)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.nll(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=tid)
    t_fit = (time.perf_counter() - t0) / 3
    eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-6, True, 0.0, trend=tid)
    eng.sweep(acq, float(y.min()), True)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.sweep(acq, float(y.min()), True)
    t = (time.perf_counter() - t0) / 3
    print("%-18s llf+grad %.2f ms   sweep %.1f ms = %.2f M candidates/s   %s" % (name, t_fit * 1e3, t * 1e3, M / t / 1e6, eng.last_timing()))
