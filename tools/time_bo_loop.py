"""Wall time of a whole ask/tell loop on the device path at the sizes of an ordinary BO run: d = 10, 20-point DoE, 200 evaluations of the
Rastrigin function; per iteration  tell = standardise + GaussianProcess.fit (multi-restart L-BFGS-B MLE, every likelihood on the GPU),
ask = EI swept over 1e5 device-generated candidates + lock-step polish of the top 32 ("sweep-device-BFGS").  The model is what
`bayes_optim.fmin` builds (`__init__.py:147-160`).  Prints the split at a few training-set sizes and the totals.
usage: python tools/time_bo_loop.py [restart_streams | bR ...]   (e.g. `1 4 8`: the MLE restarts on that many engines of the GPU at once; `b10`: restart_batch = 10, the restarts in lock step)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bogp
from bogp import optim


def main(dim=10, max_FEs=200, n_doe=20, seed=1, streams=1, batch=0, chain=False):
    f = lambda x: float(10 * len(x) + np.sum(np.asarray(x) ** 2 - 10 * np.cos(2 * np.pi * np.asarray(x))))  # noqa: E731
    lo, hi = -5.12, 5.12
    box = optim.Box([(lo, hi)] * dim)
    rng = np.random.default_rng(seed)
    np.random.seed(seed)
    X = rng.uniform(lo, hi, size=(n_doe, dim))
    y = np.array([f(x) for x in X])
    rng_len = np.full(dim, hi - lo)
    model = bogp.GaussianProcess(mean=bogp.trend.constant_trend(dim), corr="matern", thetaL=1e-3 * rng_len, thetaU=1e3 * rng_len,
                                 nugget=1e-6, optimizer="BFGS", wait_iter=3, random_start=max(10, dim), eval_budget=100 * dim,
                                 restart_streams=streams, restart_batch=batch, mle_chain_rule=chain)  # fmt: skip
    t_tell, t_ask, sizes = [], [], []
    t_all = time.perf_counter()
    while len(y) < max_FEs:
        t0 = time.perf_counter()
        ys = (y - y.mean()) / y.std()
        model.fit(X, ys.reshape(-1, 1))
        t1 = time.perf_counter()
        crit = bogp.EI(model=model, minimize=True, plugin=float(ys.min()))
        xopt, fopt = optim.argmax_restart(crit, box, eval_budget=100_000, n_restart=32, optimizer="sweep-device-BFGS")
        t2 = time.perf_counter()
        x_new = np.asarray(xopt, dtype=float)
        if np.any(np.all(np.isclose(X, x_new), axis=1)):
            x_new = rng.uniform(lo, hi, size=dim)
        X = np.vstack([X, x_new])
        y = np.append(y, f(x_new))
        t_tell.append(t1 - t0)
        t_ask.append(t2 - t1)
        sizes.append(len(y) - 1)
    total = time.perf_counter() - t_all
    t_tell, t_ask, sizes = np.array(t_tell), np.array(t_ask), np.array(sizes)
    print("== restart_streams = %d, restart_batch = %d%s" % (streams, batch, ", mle_chain_rule (extension: gradient w.r.t. log10 par)" if chain else ""))
    print("d = %d, %d evaluations (%d-point DoE), Rastrigin: best %.4f; loop wall time %.2f s = tell %.2f s + ask %.2f s (+ %.2f s of host glue)"
          % (dim, max_FEs, n_doe, y.min(), total, t_tell.sum(), t_ask.sum(), total - t_tell.sum() - t_ask.sum()))
    for n in (25, 50, 100, 150, 199):
        sel = np.abs(sizes - n) <= 3
        print("  N ~ %3d: tell (fit, %d-evaluation budget) %.1f ms, ask (1e5-candidate sweep + polish of 32 starts) %.1f ms (medians)"
              % (n, 100 * dim, 1e3 * np.median(t_tell[sel]), 1e3 * np.median(t_ask[sel])))


if __name__ == "__main__":
    for a in sys.argv[1:] or ["1"]:
        if a.startswith("c"):  # c10: restart_batch = 10 with the chain-rule gradient (an extension, not the reference's objective)
            main(batch=int(a[1:]), chain=True)
        elif a.startswith("b"):  # b10: restart_batch = 10 (the restarts in lock step, bogp_mle_batch)
            main(batch=int(a[1:]))
        else:
            main(streams=int(a))
