"""Latency of the one-point calls the reference's default inner optimiser (multi-restart L-BFGS-B, base.py:201-243) makes:
criterion(x, return_dx=True) -> predict(1 row) + gradient(1 row).  Prints microseconds per call on the current GPU."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bogp  # noqa: E402


def main():
    for N, d in ((64, 5), (512, 10), (2048, 20)):
        rng = np.random.default_rng(0)
        X = rng.uniform(-5, 5, (N, d))
        y = np.sum(X**2, axis=1)
        y = ((y - y.mean()) / y.std()).reshape(-1, 1)
        gp = bogp.GaussianProcess(corr="matern", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
        gp.set_state(np.r_[np.full(d, 0.02), 0.9], X, y)
        ei = bogp.EI(model=gp)
        x = rng.uniform(-5, 5, (1, d))
        out = {}
        for name, fn in (("predict(1 row, MSE)", lambda: gp.predict(x, eval_MSE=True)), ("gradient(1 row)", lambda: gp.gradient(x)),
                         ("EI(x)", lambda: ei(x)), ("EI(x, return_dx=True)", lambda: ei(x, return_dx=True))):  # fmt: skip
            for _ in range(20):
                fn()
            t = time.perf_counter()
            n = 300
            for _ in range(n):
                fn()
            out[name] = (time.perf_counter() - t) / n * 1e6
        print("N=%d d=%d: " % (N, d) + ", ".join("%s %.0f us" % kv for kv in out.items()))


if __name__ == "__main__":
    main()
