"""BASELINE.json configs[0] through the reference's own `fmin()`: `bogp.install()` + `bayes_optim.fmin(...)`.

Needs `bayes_optim` importable (the reference package) and an MI355X: `fmin` then builds its GaussianProcess through the
re-pointed name (`bayes_optim/__init__.py:147-160`), so the MLE, the posterior and the acquisition run on the device, while the
ask/tell loop, the DoE, the standardisation and the stopping rule are the reference's, untouched.
`examples/minimize_sphere.py` is the same problem on a hand-written ask/tell loop for boxes without the reference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bogp


def main():
    import bayes_optim  # the reference

    f = lambda x: float(np.sum(np.asarray(x) ** 2))  # noqa: E731
    # plain fmin: the reference's default inner optimiser (multi-restart L-BFGS-B, one bogp_point_eval per evaluation)
    undo = bogp.install(bayes_optim)
    xopt, fopt, n_iter, n_eval, _ = bayes_optim.fmin(f, [-5.0] * 2, [5.0] * 2, max_FEs=30, seed=42)
    undo()
    print("default BFGS       : fopt %.6g after %d evaluations" % (np.ravel(fopt)[0], n_eval))
    # the same call with the default rerouted: one sweep of 1e5 device-generated candidates + lock-step polish per ask()
    undo = bogp.install(bayes_optim, reroute_bfgs="sweep-device-BFGS", sweep_budget=100_000)
    xopt, fopt, n_iter, n_eval, _ = bayes_optim.fmin(f, [-5.0] * 2, [5.0] * 2, max_FEs=30, seed=42)
    undo()
    print("sweep-device-BFGS  : fopt %.6g after %d evaluations" % (np.ravel(fopt)[0], n_eval))


if __name__ == "__main__":
    main()
