"""`bogp.install(bayes_optim)`: route the reference's own drivers through the device engine, without editing them.

What is re-pointed (INTEGRATION.md sections 3-4; all undone by the returned callable):

  bayes_optim.base.argmax_restart               -> routed_argmax_restart    (same signature; serves the sweep family,
                                                                             hands everything else to the original)
  bayes_optim.{base,bayes_opt,extension}.AcquisitionFunction -> a per-MODEL dispatching namespace (below)
  ParallelBO._batch_arg_max_acquisition         -> fused_batch_arg_max_acquisition
  bayes_optim.GaussianProcess, bayes_optim.surrogate.GaussianProcess -> a dispatching class that builds
                                                   `bogp.GaussianProcess` (so `bayes_optim.fmin` -- which resolves the
                                                   name at call time, `__init__.py:147-160` -- runs on the device)

The contract of install() is that NOTHING the reference could do before stops working afterwards:

  * the acquisition namespace decides per model: `EI(model=<a bogp GaussianProcess>)` builds this package's criterion,
    any other model (the reference's CPU GaussianProcess, RandomForest, ...) gets the reference's own class
    (`acquisition_fun.py:107-310`); every other attribute of the namespace is the reference module's;
  * `routed_argmax_restart` serves optimizer = "sweep", "sweep-device[-lhs|-sobol]", "sweep-BFGS" (and a rerouted
    default "BFGS", see `install(reroute_bfgs=...)`) for this package's criteria; "BFGS" on a real space, constraints
    under BFGS / "OnePlusOne_Cholesky_CMA" / "MIES", non-real spaces and criteria that are not this package's all go
    to the saved original (`acquisition/optim/__init__.py:55-153`) -- our criteria are plain callables to it;
  * the GaussianProcess name builds the reference's own class for configurations the device does not serve
    (`optimizer="CMA"`, a callable correlation it does not know), with a warning.

Row f1: the reference maximises its q criteria one after the other (`bayes_opt.py:100-115`: q calls of
`_argmax_restart`, i.e. q host samplings, q uploads and q posterior passes although the criteria differ only in
t / alpha), de-duplicates afterwards (`BO.pre_eval_check`, `:27-55`) and pads what is left with random points
(`base.py:282-289`).  The fused method draws the q parameters with the reference's own sampler IN THE SAME ORDER (so
the global np.random stream advances exactly as in the reference), then makes ONE call of `optim.batch_argmax`: one
candidate design, one posterior pass, q criteria, top-k per criterion; a criterion whose best candidate is already
taken by an earlier criterion, or `np.isclose` to an evaluated point, falls back through its own top-k.  With any
other inner optimiser ("BFGS", ...) the reference's method runs unchanged.
"""
from __future__ import annotations

import types
import warnings
from copy import copy

import numpy as np

from . import acquisition, optim
from .surrogate import GaussianProcess as _DeviceGP

_SWEEPS = ("sweep",) + tuple(optim.DEVICE_DESIGNS)
_OURS = _SWEEPS + ("sweep-BFGS", "sweep-device-BFGS")  # inner optimisers only this package knows


def is_device_model(model) -> bool:
    """True for a surrogate whose posterior lives in a libbogp engine (this package's GaussianProcess or a subclass)."""
    return isinstance(model, _DeviceGP)


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
    kw = getattr(getattr(self, "_argmax_restart", None), "keywords", None) or {}  # bound by BaseBO.__set_argmax (base.py:231-243)
    if optimizer not in _SWEEPS or not is_device_model(getattr(self, "model", None)) or not optim.is_continuous(kw.get("search_space")):
        return _ORIGINAL["batch"](self, n_point, return_dx, fixed)
    if optimizer in optim.DEVICE_DESIGNS and (kw.get("h") is not None or kw.get("g") is not None or fixed):
        return _ORIGINAL["batch"](self, n_point, return_dx, fixed)  # one criterion at a time through routed_argmax_restart
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
    design = optim.DEVICE_DESIGNS.get(optimizer)
    k = int(min(32, n_point + 8))  # fall-backs: at most n_point - 1 taken by earlier criteria + a few history hits
    xs, fs = optim.batch_argmax(crits, kw["search_space"], int(budget or kw["eval_budget"]), history=_history_of(self), k=k,
                                design=design, masks=masks, values=values, h=kw.get("h"), g=kw.get("g"))  # fmt: skip
    return tuple(xs), tuple(fs)


_ORIGINAL: dict = {}
_REROUTE: dict = {}
_SURROGATE_DEFAULTS: dict = {}  # keyword defaults install() gives the device GaussianProcess (e.g. restart_batch)


def _effective(optimizer, eval_budget):
    """(optimizer, eval_budget) after `install(reroute_bfgs=...)`: the reference's DEFAULT inner optimiser for a GP on a real
    space is "BFGS" with a budget of 100 dim point evaluations (base.py:200-214, default_AQ_max_FEs) -- thousands of one-point
    device round trips per ask().  With a reroute installed those calls become one sweep of `sweep_budget` candidates."""
    if optimizer == "BFGS" and _REROUTE:
        return _REROUTE["optimizer"], _REROUTE["budget"]
    return optimizer, eval_budget


def routed_argmax_restart(obj_func, search_space, h=None, g=None, eval_budget=100, n_restart=10, wait_iter=3,
                          optimizer="BFGS", logger=None):
    """`bayes_optim.base.argmax_restart` after `install()`.

    This package's inner optimisers (the sweep family) are served by `optim.argmax_restart` when `obj_func` wraps one
    of this package's criteria over a continuous space; every other call -- the reference's "BFGS" /
    "OnePlusOne_Cholesky_CMA" / "MIES", constraints under them, mixed / integer / discrete spaces, foreign criteria or
    models -- goes to the reference's own function with the arguments untouched (optim/__init__.py:55-153), for which a
    bogp criterion is an ordinary callable.  A reroute (`install(reroute_bfgs=...)`) only ever applies to an
    unconstrained "BFGS" call on a continuous space whose criterion is this package's."""
    original = _ORIGINAL["argmax"]
    mine = optim.unwrap_criterion(obj_func)[0] is not None and optim.is_continuous(search_space)
    opt2, budget2 = optimizer, eval_budget
    if mine and optimizer == "BFGS" and h is None and g is None:
        opt2, budget2 = _effective(optimizer, eval_budget)
    if mine and opt2 in _OURS:
        return optim.argmax_restart(obj_func, search_space, h=h, g=g, eval_budget=budget2, n_restart=n_restart,
                                    wait_iter=wait_iter, optimizer=opt2, logger=logger)  # fmt: skip
    if optimizer in _OURS:  # asked for by name, but not servable: say why instead of a KeyError deep in the reference
        raise TypeError("optimizer=%r needs one of this package's criteria (a bogp.GaussianProcess as model=) on a "
                        "continuous search space" % optimizer)  # fmt: skip
    return original(obj_func, search_space, h=h, g=g, eval_budget=eval_budget, n_restart=n_restart, wait_iter=wait_iter,
                    optimizer=optimizer, logger=logger)  # fmt: skip


# ----------------------------------------------------------------------------------------------------------------------
# per-model dispatch: one name, two implementations
# ----------------------------------------------------------------------------------------------------------------------
class _Dispatch(type):
    """Metaclass of the names install() re-points: calling the class builds `_device` or `_host` (decided by
    `_pick(args, kwargs)`), `isinstance` / `issubclass` accept instances of either, attribute look-ups that the
    dispatching class does not define fall through to the device class and then the host class (so that
    `hasattr(cls, "plugin")`, bayes_opt.py:21-23, answers as both originals do)."""

    def __call__(cls, *args, **kwargs):
        if "_pick" not in cls.__dict__:
            # a user's `class My(bayes_optim.GaussianProcess)` inherits this metaclass: its overrides were written against the
            # reference's class, so it is realised on the HOST class with the subclass's own namespace on top (never silently
            # replaced by a plain device object); see _realise
            return _realise(cls)(*args, **kwargs)
        return cls._pick(args, kwargs)(*args, **kwargs)

    def __instancecheck__(cls, obj):
        if "_pick" not in cls.__dict__:  # a user's subclass: its instances are instances of its realisation (and of the realisations of ITS subclasses)
            return isinstance(obj, _realise(cls))
        return isinstance(obj, tuple(t for t in (cls._device, cls._host) if t is not None))

    def __subclasscheck__(cls, sub):
        if "_pick" not in cls.__dict__:
            return sub is cls or (isinstance(sub, _Dispatch) and cls in sub.__mro__) or issubclass(sub, _realise(cls))
        return sub is cls or issubclass(sub, tuple(t for t in (cls._device, cls._host) if t is not None))

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        for t in (cls.__dict__.get("_device"), cls.__dict__.get("_host")):
            if t is not None and hasattr(t, name):
                return getattr(t, name)
        raise AttributeError("%s has no attribute %r" % (cls.__name__, name))


_REALISED = {}


def _rebind_class_cell(v, cell):
    """`v` with the `__class__` cell of zero-argument super() re-pointed at `cell` (functions, and the functions inside static / class
    methods and properties); anything else unchanged."""
    if isinstance(v, types.FunctionType):
        if "__class__" not in v.__code__.co_freevars:
            return v
        closure = tuple(cell if n == "__class__" else c for n, c in zip(v.__code__.co_freevars, v.__closure__))
        f = types.FunctionType(v.__code__, v.__globals__, v.__name__, v.__defaults__, closure)
        f.__kwdefaults__, f.__qualname__, f.__doc__, f.__module__ = v.__kwdefaults__, v.__qualname__, v.__doc__, v.__module__
        f.__dict__.update(v.__dict__)
        f.__annotations__ = dict(getattr(v, "__annotations__", {}) or {})
        return f
    if isinstance(v, staticmethod):
        return staticmethod(_rebind_class_cell(v.__func__, cell))
    if isinstance(v, classmethod):
        return classmethod(_rebind_class_cell(v.__func__, cell))
    if isinstance(v, property):
        return property(*(None if f is None else _rebind_class_cell(f, cell) for f in (v.fget, v.fset, v.fdel)), v.__doc__)
    return v


def _bare_instance(cls):
    r = _realise(cls)
    return r.__new__(r)


def _realise(cls):
    """The ordinary class behind a user's subclass of a dispatching name: built ONCE per subclass -- so type(a) is type(b) and
    isinstance works --, on the host class (or on the realisations of the user's own intermediate subclasses), with the subclass's
    namespace on top and the `__class__` cell of its methods re-pointed at the realisation, so that zero-argument `super()` resolves
    (`super().__init__(...)`, an overriding `fit` that calls `super().fit`).  Instances pickle through the dispatching name the user's
    module holds (`__reduce_ex__`), which is the only name pickle / dill can look up."""
    r = _REALISED.get(cls)
    if r is not None:
        return r
    bases = []
    for b in cls.__bases__:
        if isinstance(b, _Dispatch):
            if "_pick" in b.__dict__:
                host = b.__dict__.get("_host")
                if host is None:
                    raise TypeError("%s derives from a dispatching bogp class that has no host class to fall back to" % cls.__name__)
                bases.append(host)
            else:
                bases.append(_realise(b))
        else:
            bases.append(b)
    cell = types.CellType()
    ns = {k: _rebind_class_cell(v, cell) for k, v in cls.__dict__.items() if k not in ("__dict__", "__weakref__")}

    def __reduce_ex__(self, protocol, _cls=cls):
        state = self.__getstate__() if hasattr(self, "__getstate__") and type(self).__getstate__ is not getattr(object, "__getstate__", None) else self.__dict__
        return (_bare_instance, (_cls,), state)

    ns.setdefault("__reduce_ex__", __reduce_ex__)
    r = type(cls.__name__, tuple(bases), ns)
    cell.cell_contents = r
    _REALISED[cls] = r
    return r


def _dispatching_criterion(name, device_cls, host_cls):
    def _pick(args, kwargs):
        model = kwargs.get("model", next((a for a in args if hasattr(a, "predict")), None))
        if device_cls is not None and (is_device_model(model) or host_cls is None):
            return device_cls
        return host_cls

    return _Dispatch(name, (), {"_device": device_cls, "_host": host_cls, "_pick": staticmethod(_pick),
                                "__doc__": "dispatches on `model`: %s on a bogp.GaussianProcess, the reference's otherwise" % name})  # fmt: skip


class _AcquisitionNamespace:
    """Stands in for the module `bayes_optim.acquisition.acquisition_fun` under the name `AcquisitionFunction`
    (`base.py:14,250,488`, `bayes_opt.py:8,21,96,187`, `extension.py:13,327`): criteria this package implements dispatch
    per model, every other attribute is the reference module's own."""

    def __init__(self, reference_module):
        self._ref = reference_module
        self._classes = {}
        for name in ("EI", "PI", "EpsilonPI", "UCB", "MGFI"):
            ours, theirs = getattr(acquisition, name, None), getattr(reference_module, name, None)
            if ours is not None:
                self._classes[name] = _dispatching_criterion(name, ours, theirs)

    def __getattr__(self, name):
        classes = self.__dict__.get("_classes", {})
        if name in classes:
            return classes[name]
        return getattr(self.__dict__["_ref"], name)

    def __dir__(self):
        return sorted(set(dir(self._ref)) | set(self._classes))


def _dispatching_surrogate(host_cls):
    def _pick(args, kwargs):
        # install(restart_batch=...): defaults of the device models the drivers build themselves (taken back if the host class is picked)
        added = [k for k in _SURROGATE_DEFAULTS if k not in kwargs]
        for k in added:
            kwargs[k] = _SURROGATE_DEFAULTS[k]
        try:  # validate on a throw-away instance: the constructor touches no device
            probe = _DeviceGP(*args, **kwargs)
            probe._trend_args()  # a trend basis the device does not evaluate is found NOW, not at fit() (it raises NotImplementedError)
            return _DeviceGP
        except NotImplementedError as e:
            for k in added:
                kwargs.pop(k, None)
            warnings.warn("bogp: this GaussianProcess configuration stays on the reference's CPU class (%s)" % e, stacklevel=3)
            return host_cls

    return _Dispatch("GaussianProcess", (), {"_device": _DeviceGP, "_host": host_cls, "_pick": staticmethod(_pick),
                                             "__doc__": "bogp.GaussianProcess where the device serves the configuration, "
                                                        "else bayes_optim's CPU class"})  # fmt: skip


def install(bayes_optim=None, fuse_batch: bool = True, reroute_bfgs: str = None, sweep_budget: int = 1_000_000, surrogate: bool = True,
            restart_batch: int = None):
    """Re-point the reference's extension points at this package (see the module docstring).  `bayes_optim` is the
    imported reference package (default: `import bayes_optim`).  Returns `uninstall()`.  Idempotent.

    `surrogate=True`: `bayes_optim.GaussianProcess` and `bayes_optim.surrogate.GaussianProcess` build
    `bogp.GaussianProcess`, hence `bayes_optim.fmin(...)` fits, predicts and maximises on the device (the defining module
    `bayes_optim.surrogate.gaussian_process` keeps the CPU class; names imported BEFORE install() keep what they had).

    `reroute_bfgs` = "sweep" | "sweep-device" | "sweep-device-lhs" | "sweep-device-sobol" | "sweep-BFGS" | "sweep-device-BFGS": drivers constructed
    WITHOUT `acquisition_optimization` fall to the reference's default "BFGS" (one device round trip per point); with a
    reroute their inner maximisation becomes one sweep of `sweep_budget` candidates instead -- no change to the driver's
    constructor call.  Constrained problems, foreign models and non-real spaces are never rerouted.

    `restart_batch` = R: device models built through the re-pointed name get `restart_batch=R` unless the call names its own."""
    if restart_batch is not None:
        # the models `fmin` / the drivers build through the re-pointed GaussianProcess name fit with their MLE restarts in lock step
        # (surrogate.GaussianProcess(restart_batch=R)): `tell()` is most of a BO loop's wall time.  Opt-in: the
        # sequential loop with scipy's optimiser stays the default, as in the reference.
        _SURROGATE_DEFAULTS["restart_batch"] = int(restart_batch)
    if reroute_bfgs is not None:
        if reroute_bfgs not in _OURS:
            raise ValueError("reroute_bfgs must be one of %s" % (_OURS,))
        _REROUTE.update(optimizer=reroute_bfgs, budget=int(sweep_budget))
    if bayes_optim is None:
        import bayes_optim  # noqa: PLC0415
    import importlib

    rbase = importlib.import_module(bayes_optim.__name__ + ".base")
    ropt = importlib.import_module(bayes_optim.__name__ + ".bayes_opt")
    racq = importlib.import_module(bayes_optim.__name__ + ".acquisition.acquisition_fun")
    rsur = importlib.import_module(bayes_optim.__name__ + ".surrogate")
    try:
        rext = importlib.import_module(bayes_optim.__name__ + ".extension")
    except Exception:  # optional module with heavier dependencies
        rext = None

    if _ORIGINAL:
        return uninstall
    holders = [m for m in (rbase, ropt, rext) if m is not None and getattr(m, "AcquisitionFunction", None) is racq]
    _ORIGINAL.update(argmax=rbase.argmax_restart, acq_holders=holders, acq=racq, batch=ropt.ParallelBO._batch_arg_max_acquisition,
                     mods=(bayes_optim, rbase, ropt, rsur), gp=(getattr(bayes_optim, "GaussianProcess", None), rsur.GaussianProcess),
                     surrogate=bool(surrogate))  # fmt: skip
    rbase.argmax_restart = routed_argmax_restart
    ns = _AcquisitionNamespace(racq)
    for m in holders:
        m.AcquisitionFunction = ns
    if fuse_batch:
        ropt.ParallelBO._batch_arg_max_acquisition = fused_batch_arg_max_acquisition
    if surrogate:
        gp = _dispatching_surrogate(rsur.GaussianProcess)
        bayes_optim.GaussianProcess = rsur.GaussianProcess = gp
    return uninstall


def uninstall():
    if not _ORIGINAL:
        return
    pkg, rbase, ropt, rsur = _ORIGINAL["mods"]
    rbase.argmax_restart = _ORIGINAL["argmax"]
    for m in _ORIGINAL["acq_holders"]:
        m.AcquisitionFunction = _ORIGINAL["acq"]
    ropt.ParallelBO._batch_arg_max_acquisition = _ORIGINAL["batch"]
    if _ORIGINAL["surrogate"]:
        pkg.GaussianProcess, rsur.GaussianProcess = _ORIGINAL["gp"]
    _ORIGINAL.clear()
    _REROUTE.clear()
    _SURROGATE_DEFAULTS.clear()
