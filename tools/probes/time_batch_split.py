"""A batch of P likelihood evaluations on ONE handle against the same evaluations as two half batches on two handles / host threads: does one half's
panel chain (small launches) hide behind the other half's whole-state updates?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bogp import _lib
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BOGP_")})
for N in (1024, 2048):
    d = 20
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1); y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2 / d), 0.9]
    engs = [_lib.Engine(0) for _ in range(2)]
    for e in engs: e.set_train(X, y)
    for P in (4, 8, 10, 16):
        pars = np.vstack([par * (1 + 0.03 * s) for s in range(P)])
        engs[0].nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        t0 = time.perf_counter()
        for _ in range(10): one = engs[0].nll_batch(2, 1, pars, 1e-6, True, 0.0, eval_grad=True)
        t1 = (time.perf_counter() - t0) / 10 * 1e6
        halves = [pars[: P // 2], pars[P // 2:]]
        res = [None, None]
        def work(i, reps):
            for _ in range(reps): res[i] = engs[i].nll_batch(2, 1, halves[i], 1e-6, True, 0.0, eval_grad=True)
        for i in range(2): work(i, 1)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, 10)) for i in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        t2 = (time.perf_counter() - t0) / 10 * 1e6
        same = np.array_equal(np.r_[res[0][0], res[1][0]], one[0])
        print("N=%d P=%d: one batch %.0f us; two half batches on two handles %.0f us (x%.2f); same bits: %s" % (N, P, t1, t2, t1 / t2, same))
    for e in engs: e.close()
