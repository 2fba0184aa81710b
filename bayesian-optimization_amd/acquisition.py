"""Acquisition functions on the GPU engine: drop-in for `bayes_optim.acquisition.acquisition_fun`.

Protocol mirrored: classes are looked up BY NAME (`base.py:485-488`), built with keywords
`model, minimize, [plugin], [t | alpha | epsilon]`, `hasattr(cls, "plugin")` decides plugin injection
(`bayes_opt.py:21-23`), `ParallelBO` reads the default `t` / `alpha` / `epsilon` off an instance
(`bayes_opt.py:96-98`), and the object is called as `criterion(X, return_dx=bool)` (`utils.py:184-195`).

Implementation is table-driven rather than a copy of the reference's class bodies:
  * values come from libbogp's acquisition kernel -- ONE device launch for any number of rows
    (closed forms + guards of `acquisition_fun.py:127-135, 153-176, 208-217, 265-290` live in csrc/kernels_acq.hip);
  * the `return_dx` chain rule (rows a12: `:139-146, 181-188, 220-227, 292-309`) is evaluated on the host from
    `model.gradient`, one row per call like the reference, through the shared `_Moments` helper.

Batched semantics (the reference raises ValueError for EI / EpsilonPI / MGFI on more than one row): row i of the
result is what the reference's single-row call returns for row i, guards included; with `return_dx=True` and several
rows the chain rule itself runs on the device (`bogp_point_eval_batch`) and the answer is (values (M, 1), dx (M, d)).
Return shapes follow the reference: one row -> shape (1,) for EI / MGFI / UCB (its Python `sum` over a (1,1) array)
and (1,1) for EpsilonPI; M rows -> (M, 1).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
from scipy.special import ndtr

from . import _lib


class norm:  # noqa: N801 -- the two members of scipy.stats.norm the chain rules use, without its argument-checking machinery
    """`scipy.stats.norm.pdf / cdf` at loc = 0, scale = 1 are exactly `exp(-x**2 / 2) / sqrt(2 pi)` and `special.ndtr(x)`
    (scipy/stats/_continuous_distns.py: `_norm_pdf`, `_norm_cdf`); calling those directly gives the same bits at a tenth
    of the host time -- the reference's BFGS loop evaluates them once per point."""

    _C = np.sqrt(2 * np.pi)

    @staticmethod
    def pdf(x):
        return np.exp(-np.asarray(x) ** 2 / 2.0) / norm._C

    @staticmethod
    def cdf(x):
        return ndtr(x)


class _PositiveParameter:
    """Data descriptor for a strictly positive criterion parameter (the reference asserts `> 0` in every setter:
    alpha `:123-125`, epsilon `:204-206`, t `:259-263`), with an optional transform (MGFI clamps t at 22.36)."""

    def __init__(self, transform: Optional[Callable[[float], float]] = None, zero_ok_attr: Optional[str] = None):
        self.transform = transform
        self.zero_ok_attr = zero_ok_attr

    def __set_name__(self, owner, name):
        self.slot = "_" + name

    def __get__(self, obj, objtype=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        zero_ok = bool(self.zero_ok_attr and getattr(obj, self.zero_ok_attr, False))
        assert value > 0 or (zero_ok and value == 0)
        setattr(obj, self.slot, self.transform(value) if self.transform else value)


class _Moments:
    """Posterior moments and their input-gradients at ONE row, in the orientation the criteria use:
    y (sign-flipped when maximising, `:52-64`), sd = sqrt(MSE), dy and dsd as (1, d) rows (`:66-80`)."""

    def __init__(self, criterion: "AcquisitionFunction", X: np.ndarray, moments=None):
        model, sign = criterion.model, (1.0 if criterion.minimize else -1.0)
        if moments is None:
            mu, mse = model.predict(X, eval_MSE=True)
            dmu, dmse = model.gradient(np.array(X, dtype=float))
        else:  # already computed by the fused one-point device call
            mu, mse, dmu, dmse = moments
        self.y = sign * mu
        self.sd = np.sqrt(mse)
        self.dy = sign * dmu.T
        with np.errstate(all="ignore"):
            self.dsd = dmse.T / (2.0 * self.sd)
        self.d = X.shape[1]

    def zeros(self):
        return np.zeros((self.d, 1))  # the shape of the reference's guard / exception fall-backs


class AcquisitionFunction:
    """Common machinery; concrete criteria set `acq_id` and override `_par` / `_dx`."""

    acq_id: int = -1

    def __init__(self, model=None, minimize: bool = True):
        self.model = model
        self.minimize = minimize

    @property
    def model(self):
        return self._model

    @model.setter
    def model(self, model):
        if model is None:
            raise ValueError("model cannot be None")
        assert hasattr(model, "predict")
        self._model = model

    # -- what the sweep needs from a criterion ----------------------------------------------------------
    def acq_par(self) -> float:
        return 0.0

    def effective_plugin(self) -> float:
        return 0.0

    def check_X(self, X) -> np.ndarray:
        return np.atleast_2d(np.asarray(X, dtype=float))

    # -- evaluation ---------------------------------------------------------------------------------------
    def _values(self, X: np.ndarray) -> np.ndarray:
        """Posterior + this criterion for all rows of X in one device call (bogp_sweep with acq_out)."""
        if getattr(self._model, "_committed_par", None) is None:
            raise Exception("The model is not fitted yet!")
        eng = self._model.engine
        eng.upload_candidates(self._model._check_X(X))
        _, _, vals = eng.sweep([(self.acq_id, self.acq_par())], self.effective_plugin(), self.minimize, return_values=True)
        return vals[0]

    _single_row_shape = (1,)

    def _fused_point(self, X: np.ndarray):
        """value + moments + their gradients at one row in ONE device call (bogp_point_eval), or None when the model
        needs the separate predict / gradient entry points (polynomial trend basis, several targets, foreign model)."""
        model = self._model
        if getattr(model, "_committed_par", None) is None:
            raise Exception("The model is not fitted yet!")
        eng = getattr(model, "engine", None)
        fused = getattr(model, "_fused_point_ok", None)
        if eng is None or fused is None or not hasattr(eng, "point_eval") or not fused():
            return None
        x = model._check_X(X)
        mu, mse, dmu, dmse, vals = eng.point_eval(x[0], [(self.acq_id, self.acq_par())], self.effective_plugin(), self.minimize)
        mom = (np.array([[mu]]), np.array([[mse]]), dmu.reshape(-1, 1), dmse.reshape(-1, 1))
        return vals, mom

    def __call__(self, X, return_dx: bool = False):
        X = self.check_X(X)
        if X.shape[0] == 1:  # the one-point call of the reference's inner optimisers: one device round trip
            fused = self._fused_point(X)
            if fused is not None:
                value = fused[0].reshape(self._single_row_shape)
                return self._dx(_Moments(self, X, fused[1]), value) if return_dx else value
        if return_dx and X.shape[0] != 1:
            # the reference stops here ("x must be a vector!", gpr.py:548-549).  Row i of the answer below is what its
            # one-row call returns for row i: (values (M, 1), gradients (M, d)), one device round trip for all rows
            # (bogp_point_eval_batch evaluates the chain rule of `_dx` on the device)
            model, eng = self._model, getattr(self._model, "engine", None)
            fused = getattr(model, "_fused_point_ok", None)
            if eng is None or fused is None or not hasattr(eng, "point_eval_batch") or not fused():
                raise Exception("x must be a vector!")
            if getattr(model, "_committed_par", None) is None:
                raise Exception("The model is not fitted yet!")
            _, _, _, _, vals, dvals = eng.point_eval_batch(model._check_X(X), [(self.acq_id, self.acq_par())], self.effective_plugin(), self.minimize)
            return vals.reshape(-1, 1), dvals[:, 0, :]
        v = self._values(X)
        value = v.reshape(self._single_row_shape) if X.shape[0] == 1 else v.reshape(-1, 1)
        if not return_dx:
            return value
        return self._dx(_Moments(self, X), value)

    def _dx(self, m: _Moments, value):
        raise NotImplementedError


class ImprovementBased(AcquisitionFunction):
    """Criteria measured against a plug-in value: best observed fitness unless given (`:87-104`).
    The stored value is already oriented for minimisation (negated when maximising)."""

    def __init__(self, plugin: float = None, **kwargs):
        super().__init__(**kwargs)
        self.plugin = plugin

    @property
    def plugin(self):
        return self._plugin

    @plugin.setter
    def plugin(self, value):
        sign = 1.0 if self.minimize else -1.0
        if value is not None:
            self._plugin = sign * value
        elif hasattr(self._model, "y"):
            y = self._model.y
            self._plugin = np.min(y) if self.minimize else sign * np.max(y)
        else:
            self._plugin = None

    def effective_plugin(self) -> float:
        if self._plugin is None:  # the model had no data when the criterion was built: resolve now
            self.plugin = None
        return float(self._plugin)


class UCB(AcquisitionFunction):
    """y + alpha sd, maximised as-is even when minimising (`:107-147`)."""

    acq_id = _lib.ACQ_UCB
    alpha = _PositiveParameter()

    def __init__(self, alpha: float = 0.5, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha

    def acq_par(self):
        return float(self.alpha)

    def _dx(self, m, value):
        return value, m.dy + self.alpha * m.dsd


class EI(ImprovementBased):
    """Expected improvement (`:150-189`)."""

    acq_id = _lib.ACQ_EI

    def _dx(self, m, value):
        if m.sd / np.sqrt(self._model.sigma2) < 1e-6:  # the small-variance guard returns (0, zeros)
            return 0, m.zeros()
        z = (self.plugin - m.y) / m.sd
        return value, norm.pdf(z) * m.dsd - norm.cdf(z) * m.dy


class EpsilonPI(ImprovementBased):
    """epsilon-probability of improvement (`:192-228`)."""

    acq_id = _lib.ACQ_EPSILON_PI
    _allow_zero = False
    _single_row_shape = (1, 1)
    epsilon = _PositiveParameter(zero_ok_attr="_allow_zero")

    def __init__(self, epsilon=1e-10, **kwargs):
        super().__init__(**kwargs)
        self.epsilon = epsilon

    def acq_par(self):
        return float(self.epsilon)

    def _dx(self, m, value):
        shrink = (1 - self.epsilon) if m.y > 0 else (1 + self.epsilon)
        with np.errstate(all="ignore"):
            z = (self._plugin - shrink * m.y) / m.sd
            slope = -(shrink * m.dy + z * m.dsd) * norm.pdf(z) / m.sd
        return value, slope


class PI(EpsilonPI):
    """Probability of improvement = EpsilonPI with epsilon = 0.

    The reference's PI cannot be constructed: it forwards epsilon=0 into a setter that asserts eps > 0
    (`:204-206, 232-235`).  Here epsilon = 0 is admitted for this subclass only."""

    _allow_zero = True

    def __init__(self, **kwargs):
        kwargs["epsilon"] = 0
        super().__init__(**kwargs)


class MGFI(ImprovementBased):
    """Moment-generating function of the improvement (`:238-310`); t is clamped at 22.36 against overflow."""

    acq_id = _lib.ACQ_MGFI
    t = _PositiveParameter(transform=lambda t: min(t, 22.36))

    def __init__(self, t: float = 1, **kwargs):
        super().__init__(**kwargs)
        self.t = t

    def acq_par(self):
        return float(self.t)

    def _dx(self, m, value):
        if np.isclose(m.sd, 0):
            return np.array([0.0]), m.zeros()
        t, plugin, var = self.t, self._plugin, m.sd**2.0
        with np.errstate(all="raise"):
            try:
                z = (plugin - (m.y - t * var)) / m.sd
                scale = np.exp(t * (plugin + t * var / 2 - m.y - 1))
                dz = -((m.dy - 2.0 * t * m.sd * m.dsd) + z * m.dsd) / m.sd
                slope = scale * (norm.pdf(z) * dz + norm.cdf(z) * (t**2 * m.sd * m.dsd - t * m.dy))
            except FloatingPointError:  # the reference turns warnings into a zero gradient
                slope = m.zeros()
        return value, slope
