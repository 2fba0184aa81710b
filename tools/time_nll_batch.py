"""Likelihood + gradient evaluations per second: P parameter vectors per bogp_nll_batch call against P sequential bogp_nll calls
(gpr.py:920-1040; the restarts of the MLE, gpr.py:1127-1162, evaluated together).  usage: python tools/time_nll_batch.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from bogp import _lib


def main(sizes):
    eng = _lib.Engine(0)
    d = 10
    for N in sizes:
        rng = np.random.default_rng(N)
        X = rng.uniform(-5, 5, size=(N, d))
        y = np.sum(X**2, axis=1)
        y = ((y - y.mean()) / y.std()).reshape(-1, 1)
        eng.set_train(X, y)
        kern, mode, nv = _lib.KERNEL_MATERN32, _lib.MODE_NOISY, 1e-6
        base = np.r_[np.full(d, 0.05), 0.9]
        reps = 200 if N <= 256 else (40 if N <= 1024 else 10)
        for _ in range(3):
            eng.nll(kern, mode, base, nv, True, 0.0, eval_grad=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.nll(kern, mode, base, nv, True, 0.0, eval_grad=True)
        t_seq = (time.perf_counter() - t0) / reps
        line = "N = %4d, d = %d (path %d): sequential %7.1f us / evaluation (%6.0f /s)" % (N, d, _lib.load().bogp_nll_path(N, d, 0, 1), 1e6 * t_seq, 1 / t_seq)
        print(line)
        only = [int(v) for v in os.environ.get("BOGP_TIME_P", "").split(",") if v]  # e.g. BOGP_TIME_P=16 under rocprofv3
        for P in only or (1, 2, 4, 8, 10, 16, 32, 64):
            if N > 1024 and P > 16:
                continue
            pars = np.tile(base, (P, 1)) * 10.0 ** rng.uniform(-0.3, 0.3, size=(P, d + 1))
            for _ in range(3):
                eng.nll_batch(kern, mode, pars, nv, True, 0.0, eval_grad=True)
            r = max(3, reps // 2)
            t0 = time.perf_counter()
            for _ in range(r):
                eng.nll_batch(kern, mode, pars, nv, True, 0.0, eval_grad=True)
            t = (time.perf_counter() - t0) / r
            print("    P = %2d: %8.1f us / batch = %7.1f us / evaluation (%7.0f /s, %.1fx)" % (P, 1e6 * t, 1e6 * t / P, P / t, t_seq * P / t))
    eng.close()


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [40, 100, 150, 200, 256, 512, 1024, 2048])
