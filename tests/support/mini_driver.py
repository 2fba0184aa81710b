"""TEST-ONLY minimal stand-in for the attribute surface of `bayes_optim.ParallelBO` that
`bogp.integration.fused_batch_arg_max_acquisition` touches (`bayes_opt.py:58-115`, `base.py:231-243, 482-494`).
The GPU box has no reference tree, so the `-m gpu` half of the f1 test drives the device classes through this; the
other half (tests/test_dropin_reference.py, build container) runs the REAL ParallelBO on the oracle-backed engine."""
import functools

import numpy as np

import bogp
from bogp import optim


class MiniParallelBO:
    def __init__(self, model, bounds, acquisition_fun="MGFI", acquisition_par=None, optimizer="sweep", max_FEs=4096, seed=0,
                 history=None):
        self.model, self.minimize, self.dim = model, True, len(bounds)
        self._acquisition_fun = acquisition_fun
        self._acquisition_par = dict(acquisition_par or {"t": 2})
        self._optimizer = optimizer
        self.logger = None
        self.search_space = optim.Box(bounds, random_seed=seed)
        if history is not None:
            self.data = np.asarray(history, dtype=float)
        if acquisition_fun == "MGFI":  # the two samplers of bayes_opt.py:81-90
            self._par_name = "t"
            self._sampler = lambda x: np.exp(np.log(x["t"]) + 0.5 * np.random.randn())
        else:
            self._par_name = "alpha"
            self._sampler = lambda x: 1 / (1 + np.exp((x["alpha"] * 4 - 2) + 0.6 * np.random.randn()))
        self._argmax_restart = functools.partial(optim.argmax_restart, search_space=self.search_space, h=None, g=None,
                                                 eval_budget=int(max_FEs), n_restart=2, wait_iter=3, optimizer=optimizer)  # fmt: skip

    def _create_acquisition(self, fun=None, par=None, return_dx=False, fixed=None):
        par = dict(par or self._acquisition_par)
        par.update(model=self.model, minimize=self.minimize)
        cls = getattr(bogp.acquisition, fun or self._acquisition_fun)
        if hasattr(cls, "plugin"):
            par.setdefault("plugin", float(np.min(self.model.y)))
        return functools.partial(cls(**par), return_dx=return_dx)
