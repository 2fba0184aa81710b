"""ask() wall time through the ParallelBO hook (BASELINE metric: "... and ask() wall-time"): what
`bogp.install()` makes `ParallelBO.ask()` spend in `_batch_arg_max_acquisition` for n_point = 1 / 8 proposals over 1e6
candidates at the C3 model size (N = 2048, d = 20, Matern-5/2, MGFI), with host-sampled candidates ("sweep": numpy
sampling + 160 MB H2D, the reference's own data path) and device-generated ones ("sweep-device"), plus tell(): one
complete `GaussianProcess.fit` (multi-restart L-BFGS-B, every evaluation on the device) at the same size.
The driver surface is tests/support/mini_driver.py (the reference tree does not travel to the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bogp  # noqa: E402
from bogp import integration  # noqa: E402
from support.mini_driver import MiniParallelBO  # noqa: E402

N, d, M = 2048, 20, 1_000_000
rng = np.random.default_rng(0)
X = rng.uniform(-5, 5, size=(N, d))
y = np.sum(X**2, axis=1)
y = (y - y.mean()) / y.std()
gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-4] * d, thetaU=[1e0] * d, nugget=1e-6,
                          random_start=5, wait_iter=3, eval_budget=400)
for rep in range(2):  # the first call in a process pays the library / rocBLAS initialisation
    np.random.seed(0)
    t0 = time.perf_counter()
    gp.fit(X, y)
    t_fit = time.perf_counter() - t0
    print("tell(): GaussianProcess.fit at N=%d d=%d, %s call: %.3f s (llf %.3f, budget %d likelihood evaluations)"
          % (N, d, "first" if rep == 0 else "second", t_fit, gp.log_likelihood_, gp.eval_budget))
for opt in ("sweep", "sweep-device"):
    for q in (1, 8):
        drv = MiniParallelBO(gp, [(-5, 5)] * d, "MGFI", {"t": 2}, opt, M, seed=3, history=X[:64])
        np.random.seed(1)
        integration.fused_batch_arg_max_acquisition(drv, q, False)  # warm-up
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            xs, fs = integration.fused_batch_arg_max_acquisition(drv, q, False)
            ts.append(time.perf_counter() - t0)
        assert len(xs) == q and len({tuple(x) for x in xs}) == q
        print("ask(): optimizer=%-12s n_point=%d  M=%d: %.1f ms (min of 3; %s)" % (opt, q, M, min(ts) * 1e3, ", ".join("%.1f" % (t * 1e3) for t in ts)))

# r03: the single-proposal inner maximisers behind `argmax_restart`'s signature, as BO.ask() (n_point = 1) reaches them:
#   "BFGS"        the reference's DEFAULT for a GP on a real space (base.py:200-214): 100 d = 2000 one-point evaluations of
#                 EI(x, return_dx=True), each one bogp_point_eval round trip
#   "sweep-BFGS"  one sweep of M candidates, its top-32 polished in lock step on the device (bogp_polish)
ei = bogp.EI(model=gp, minimize=True)
box = bogp.optim.Box([(-5.0, 5.0)] * d, random_seed=5)
for opt, kw in (("BFGS", dict(eval_budget=100 * d, n_restart=10, wait_iter=3)), ("sweep-BFGS", dict(eval_budget=M, n_restart=32)),
                ("sweep-device-BFGS", dict(eval_budget=M, n_restart=32)), ("sweep-device-BFGS", dict(eval_budget=100_000, n_restart=32)),
                ("sweep-device", dict(eval_budget=M))):
    np.random.seed(2)
    bogp.argmax_restart(ei, box, optimizer=opt, **kw)  # warm-up
    ts, fs = [], []
    for rep in range(3):
        t0 = time.perf_counter()
        x, f = bogp.argmax_restart(ei, box, optimizer=opt, **kw)
        ts.append(time.perf_counter() - t0)
        fs.append(f)
    print("ask(): argmax_restart(EI, optimizer=%-10s %s): %.1f ms (min of 3; best EI found %.4g)" % (opt + ",", ", ".join("%s=%s" % kv for kv in kw.items()), min(ts) * 1e3, max(fs)))
