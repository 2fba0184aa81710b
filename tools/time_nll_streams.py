"""Throughput of likelihood + gradient evaluations issued from T host threads on T engines (= T HIP streams) of ONE GPU: how far the
chains of small launches of concurrent MLE restarts overlap (GaussianProcess(restart_streams=T)).  usage: python tools/time_nll_streams.py [N d]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bogp import _lib

N, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (150, 10)
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d))
y = np.sum(X**2, axis=1)
y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
par = np.r_[np.full(d, 0.2 / d), 0.9]
engs = [_lib.Engine(0) for _ in range(8)]
for e in engs:
    e.set_train(X, y)
    e.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)
n = 300


def work(e):
    for _ in range(n):
        e.nll(2, 1, par, 1e-6, True, 0.0, eval_grad=True)


print("N = %d, d = %d: likelihood + gradient evaluations per second from T threads / engines / streams" % (N, d))
for T in (1, 2, 4, 8):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as pool:
        list(pool.map(work, engs[:T]))
    dt = time.perf_counter() - t0
    print("  T = %d: %7.0f evaluations/s  (%.0f us per evaluation per thread)" % (T, T * n / dt, dt / n * 1e6))
