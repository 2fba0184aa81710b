"""`bogp.install(bayes_optim)`: route the reference's own drivers through the device engine, without editing them.

Three module attributes and one method are re-pointed (INTEGRATION.md sections 3-4; all undone by the returned callable):

  bayes_optim.base.argmax_restart              -> bogp.argmax_restart      (same signature; adds optimizer="sweep" ...)
  bayes_optim.base.AcquisitionFunction         -> bogp.acquisition         (classes looked up by name, base.py:485-488)
  bayes_optim.bayes_opt.AcquisitionFunction    -> bogp.acquisition         (`hasattr(cls, "plugin")`, bayes_opt.py:21-23)
  ParallelBO._batch_arg_max_acquisition        -> fused_batch_arg_max_acquisition (below)

The last one is SURVEY.md row f1.  The reference maximises its q criteria one after the other (`bayes_opt.py:100-115`:
q calls of `_argmax_restart`, i.e. q host samplings, q uploads and q posterior passes although the criteria differ only
in t / alpha), de-duplicates afterwards (`BO.pre_eval_check`, `:27-55`) and pads what is left with random points
(`base.py:282-289`).  The fused method draws the q parameters with the reference's own sampler IN THE SAME ORDER (so the
global np.random stream advances exactly as in the reference), then makes ONE call of `optim.batch_argmax`: one candidate
design, one posterior pass, q criteria, top-k per criterion; a criterion whose best candidate is already taken by an
earlier criterion, or `np.isclose` to an evaluated point, falls back through its own top-k.  With any other inner
optimiser ("BFGS", ...) the reference's method runs unchanged.
"""
from __future__ import annotations

from copy import copy

import numpy as np

from . import acquisition, optim

_SWEEPS = ("sweep",) + tuple(optim.DEVICE_DESIGNS)


def _history_of(bo):
    """Evaluated points as a float (n, dim) array in the search space's variable order (what pre_eval_check compares
    with np.isclose, bayes_opt.py:41-48), or None before the first tell()."""
    data = getattr(bo, "data", None)
    if data is None or len(data) == 0:
        return None
    return np.asarray(np.asarray(data)[:, : bo.dim], dtype=float)


def fused_batch_arg_max_acquisition(self, n_point: int, return_dx: bool, fixed=None):
    """Drop-in body for `ParallelBO._batch_arg_max_acquisition` (bayes_opt.py:100-115): same arguments, same
    `(candidates, values)` return (two q-tuples)."""
    optimizer, budget = _effective(getattr(self, "_optimizer", None), None)
    if optimizer not in _SWEEPS:
        return _ORIGINAL["batch"](self, n_point, return_dx, fixed)
    wrapped = []
    for _ in range(n_point):  # bayes_opt.py:101-106 verbatim in effect: same draws, same order
        _par = self._sampler(self._acquisition_par)
        _acquisition_par = copy(self._acquisition_par)
        _acquisition_par.update({self._par_name: _par})
        wrapped.append(self._create_acquisition(par=_acquisition_par, return_dx=return_dx, fixed=fixed))
    crits, masks, values = [], None, None
    for w in wrapped:
        c, m, v = optim.unwrap_criterion(w)
        if c is None:  # not one of this package's criteria: nothing to fuse
            return tuple(zip(*[list(self._argmax_restart(w, logger=self.logger)) for w in wrapped]))
        crits.append(c)
        masks, values = m, v
    kw = self._argmax_restart.keywords  # bound by BaseBO.__set_argmax (base.py:231-243)
    if kw.get("h") is not None or kw.get("g") is not None:
        raise NotImplementedError("constraints are handled by the reference's penalised optimisers, not the sweep")
    design = optim.DEVICE_DESIGNS.get(optimizer)
    k = int(min(32, n_point + 8))  # fall-backs: at most n_point - 1 taken by earlier criteria + a few history hits
    xs, fs = optim.batch_argmax(crits, kw["search_space"], int(budget or kw["eval_budget"]), history=_history_of(self), k=k,
                                design=design, masks=masks, values=values)  # fmt: skip
    return tuple(xs), tuple(fs)


_ORIGINAL: dict = {}
_REROUTE: dict = {}


def _effective(optimizer, eval_budget):
    """(optimizer, eval_budget) after `install(reroute_bfgs=...)`: the reference's DEFAULT inner optimiser for a GP on a real
    space is "BFGS" with a budget of 100 dim point evaluations (base.py:200-214, default_AQ_max_FEs) -- thousands of one-point
    device round trips per ask().  With a reroute installed those calls become one sweep of `sweep_budget` candidates."""
    if optimizer == "BFGS" and _REROUTE:
        return _REROUTE["optimizer"], _REROUTE["budget"]
    return optimizer, eval_budget


def _rerouting_argmax_restart(obj_func, search_space, h=None, g=None, eval_budget=100, n_restart=10, wait_iter=3,
                              optimizer="BFGS", logger=None):
    """`optim.argmax_restart` behind the reroute of `install(reroute_bfgs=...)`: constrained problems and criteria that are not
    this package's keep the optimiser the caller asked for."""
    opt2, budget2 = _effective(optimizer, eval_budget)
    if opt2 != optimizer and (h is not None or g is not None or optim.unwrap_criterion(obj_func)[0] is None):
        opt2, budget2 = optimizer, eval_budget
    return optim.argmax_restart(obj_func, search_space, h=h, g=g, eval_budget=budget2, n_restart=n_restart, wait_iter=wait_iter,
                                optimizer=opt2, logger=logger)


def install(bayes_optim=None, fuse_batch: bool = True, reroute_bfgs: str = None, sweep_budget: int = 1_000_000):
    """Re-point the reference's extension points at this package (see the module docstring).  `bayes_optim` is the
    imported reference package (default: `import bayes_optim`).  Returns `uninstall()`.  Idempotent.
    `reroute_bfgs` = "sweep" | "sweep-device" | "sweep-device-lhs" | "sweep-device-sobol" | "sweep-BFGS": drivers constructed
    WITHOUT `acquisition_optimization` fall to the reference's default "BFGS" (one device round trip per point); with a
    reroute their inner maximisation becomes one sweep of `sweep_budget` candidates instead -- no change to the driver's
    constructor call."""
    if reroute_bfgs is not None:
        if reroute_bfgs not in _SWEEPS + ("sweep-BFGS",):
            raise ValueError("reroute_bfgs must be one of %s" % (_SWEEPS + ("sweep-BFGS",),))
        _REROUTE.update(optimizer=reroute_bfgs, budget=int(sweep_budget))
    if bayes_optim is None:
        import bayes_optim  # noqa: PLC0415
    import bayes_optim.base as rbase
    import bayes_optim.bayes_opt as ropt

    if _ORIGINAL:
        return uninstall
    _ORIGINAL.update(argmax=rbase.argmax_restart, acq_base=rbase.AcquisitionFunction, acq_opt=ropt.AcquisitionFunction,
                     batch=ropt.ParallelBO._batch_arg_max_acquisition, mods=(rbase, ropt))  # fmt: skip
    rbase.argmax_restart = _rerouting_argmax_restart
    rbase.AcquisitionFunction = ropt.AcquisitionFunction = acquisition
    if fuse_batch:
        ropt.ParallelBO._batch_arg_max_acquisition = fused_batch_arg_max_acquisition
    return uninstall


def uninstall():
    if not _ORIGINAL:
        return
    rbase, ropt = _ORIGINAL["mods"]
    rbase.argmax_restart = _ORIGINAL["argmax"]
    rbase.AcquisitionFunction, ropt.AcquisitionFunction = _ORIGINAL["acq_base"], _ORIGINAL["acq_opt"]
    ropt.ParallelBO._batch_arg_max_acquisition = _ORIGINAL["batch"]
    _ORIGINAL.clear()
    _REROUTE.clear()
