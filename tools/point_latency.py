"""Latency of the one-point call and of the batched / polished flavours at C3 size (N = 2048, d = 20, Matern-5/2):
  python tools/point_latency.py > profiles/r03_point_call_latency.txt      (on the GPU box)
VERDICT r02 item 4 asks for <= 60 us per `EI(x, return_dx=True)`; item 3 for a 32-start polish in <= 20 ms."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bogp  # noqa: E402
from bogp import _lib  # noqa: E402


def timeit(f, n, warm=20):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    for N, d, kernel, name in ((2048, 20, _lib.KERNEL_MATERN52, "C3: N=2048 d=20 Matern-5/2"), (512, 10, _lib.KERNEL_SE, "C2: N=512 d=10 SE"),
                               (128, 5, _lib.KERNEL_MATERN32, "N=128 d=5 Matern-3/2"), (8192, 50, _lib.KERNEL_SE, "C5: N=8192 d=50 SE")):  # fmt: skip
        rng = np.random.default_rng(0)
        X = rng.uniform(-5, 5, size=(N, d))
        y = np.sum(X**2, axis=1)
        y = ((y - y.mean()) / y.std()).reshape(-1, 1)
        par = np.r_[np.full(d, 0.2 / d), 0.9]
        eng = _lib.Engine(0)
        eng.set_train(X, y)
        eng.commit(kernel, _lib.MODE_NOISY, par, 1e-6, False, 0.0)
        pl = float(y.min())
        x = rng.uniform(-5, 5, size=d)
        acq = [(_lib.ACQ_EI, 0.0)]
        print("== %s" % name)
        print("engine.point_eval (1 point, EI, value + moments + gradients): %8.1f us" % timeit(lambda: eng.point_eval(x, acq, pl, True), 2000))
        print("engine.gradient   (1 point)                                 : %8.1f us" % timeit(lambda: eng.gradient(x), 2000))
        gp = bogp.GaussianProcess(corr="matern52" if kernel == _lib.KERNEL_MATERN52 else ("matern" if kernel == _lib.KERNEL_MATERN32 else "squared_exponential"),
                                  thetaL=[1e-5] * d, thetaU=[1e2] * d, nugget=1e-6)  # fmt: skip
        gp._engine = eng
        gp.set_state(par, X, y)
        ei = bogp.EI(model=gp, minimize=True)
        x1 = x[None, :]
        print("bogp.EI(x, return_dx=True) (the reference's BFGS call)       : %8.1f us" % timeit(lambda: ei(x1, return_dx=True), 2000))
        for B in (8, 32, 128, 512):
            Xb = rng.uniform(-5, 5, size=(B, d))
            t = timeit(lambda: eng.point_eval_batch(Xb, acq, pl, True), 300)
            print("engine.point_eval_batch B = %3d                               : %8.1f us  (%.2f us per point)" % (B, t, t / B))
        if d <= 64:
            Xs = rng.uniform(-5, 5, size=(100_000, d))
            eng.upload_candidates(Xs)
            tv, ti = eng.sweep_topk(acq, pl, True, 32)
            starts = Xs[ti[0]]
            lo, hi = np.full(d, -5.0), np.full(d, 5.0)
            eng.polish(starts, lo, hi, acq[0], pl, True, max_evals=50)
            t0 = time.perf_counter()
            Xp, fp, ne = eng.polish(starts, lo, hi, acq[0], pl, True, max_evals=50)
            dt = (time.perf_counter() - t0) * 1e3
            print("engine.polish 32 starts, <= 50 evaluations each               : %8.2f ms  (evaluations used: min %d median %d max %d; "
                  "best %.6g -> %.6g)" % (dt, ne.min(), int(np.median(ne)), ne.max(), tv[0].max(), fp.max()))
        eng.close()


if __name__ == "__main__":
    main()
