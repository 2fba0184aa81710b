"""Sweep time against the number of candidates M and training points N (Matern-5/2, EI, candidates generated on the
device): where the fused argmax sweep is latency bound and where it reaches the MFMA roof.  Prints one line per (N, M)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bogp import _lib  # noqa: E402


def main():
    eng = _lib.Engine(0)
    print("%6s %4s %9s %10s %14s %8s %8s   %s" % ("N", "d", "M", "ms/sweep", "candidates/s", "TFLOP/s", "of peak", "kernels of the last sweep (ms): producer / contraction / acquisition+argmax"))
    sizes = ((128, 5), (256, 10), (384, 10), (512, 10), (768, 10), (1024, 10), (1536, 20), (2048, 20), (8192, 50))
    if len(sys.argv) > 1:
        sizes = tuple((int(a), 10 if int(a) < 1536 else 20) for a in sys.argv[1:])
    for N, d in sizes:
        rng = np.random.default_rng(0)
        X = rng.uniform(-5, 5, size=(N, d))
        y = np.sum(X**2, axis=1)
        y = ((y - y.mean()) / y.std()).reshape(-1, 1)
        eng.set_train(X, y)
        eng.commit(3, 1, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6)
        for M in (1_000, 10_000, 100_000, 1_000_000):
            if N == 8192 and M > 100_000:
                continue
            eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=1)
            reps = 3 if M >= 100_000 else 20
            eng.sweep([(0, 0.0)], float(y.min()), True)
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.sweep([(0, 0.0)], float(y.min()), True)
            ms = (time.perf_counter() - t0) / reps * 1e3
            lt = eng.last_timing()
            tf = (float(N) * N + 3.0 * N) * M / ms * 1e3 / 1e12
            print("%6d %4d %9d %10.3f %14.3e %8.2f %8.3f   %.3f / %.3f / %.3f" % (N, d, M, ms, M / ms * 1e3, tf, tf / 78.6, lt["corr_ms"], lt["contract_ms"], lt["acquisition_ms"]))


if __name__ == "__main__":
    main()
