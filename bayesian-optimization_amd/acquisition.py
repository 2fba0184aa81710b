"""Acquisition functions on the GPU engine: drop-in for `bayes_optim.acquisition.acquisition_fun`.

Same class names, constructor keywords, properties and `__call__(X, return_dx)` contract as the reference
(`acquisition/acquisition_fun.py:22-310`); `BaseBO._create_acquisition` finds them by name (`base.py:485-488`) and
`hasattr(cls, "plugin")` decides plugin injection (`bayes_opt.py:21-23`).  Values come from libbogp's acquisition
kernel (one launch for any number of rows); the `return_dx` chain rule (a12) runs on the host from
`model.gradient`, one row at a time like the reference.

Batched semantics (the reference raises ValueError for EI / EpsilonPI / MGFI on more than one row): row i of the
result is what the reference's single-row call returns for row i, guards included.
Return shapes follow the reference: one row -> shape (1,) for EI / MGFI / UCB (Python `sum` over a (1,1) array) and
(1,1) for EpsilonPI; M rows -> (M, 1).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Tuple

import numpy as np
from scipy.stats import norm

from . import _lib


class AcquisitionFunction(ABC):
    """acquisition_fun.py:22-84."""

    acq_id = None  # libbogp criterion id

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
        self._model = model
        assert hasattr(self._model, "predict")

    def acq_par(self) -> float:
        return 0.0

    def effective_plugin(self) -> float:
        return 0.0

    @abstractmethod
    def __call__(self, X, return_dx: bool = False):
        raise NotImplementedError

    def check_X(self, X) -> np.ndarray:
        return np.atleast_2d(np.asarray(X, dtype=float))

    # -- device evaluation --------------------------------------------------------------------------
    def _values(self, X: np.ndarray) -> np.ndarray:
        """All rows in one device call: posterior + this criterion (libbogp bogp_sweep with acq_out)."""
        eng = self._model.engine
        if getattr(self._model, "_committed_par", None) is None:
            raise Exception("The model is not fitted yet!")
        X = self._model._check_X(X)
        eng.upload_candidates(X)
        _, _, vals = eng.sweep([(self.acq_id, self.acq_par())], self.effective_plugin(), self.minimize, return_values=True)
        return vals[0]

    def _predict(self, X) -> Tuple[np.ndarray, np.ndarray]:
        """acquisition_fun.py:52-64."""
        y_hat, sd2 = self._model.predict(X, eval_MSE=True)
        if not self.minimize:
            y_hat = -1 * y_hat
        return y_hat, np.sqrt(sd2)

    def _gradient(self, X) -> Tuple[np.ndarray, np.ndarray]:
        """acquisition_fun.py:66-80: (1, d) rows."""
        y_dx, sd2_dx = self._model.gradient(np.array(X, dtype=float))
        if not self.minimize:
            y_dx = -1.0 * y_dx
        return y_dx.T, sd2_dx.T

    def _shape(self, v: np.ndarray, n_sample: int):
        return v.reshape(1) if n_sample == 1 else v.reshape(-1, 1)


class ImprovementBased(AcquisitionFunction):
    """acquisition_fun.py:87-104."""

    def __init__(self, plugin: float = None, **kwargs):
        super().__init__(**kwargs)
        self.plugin = plugin

    @property
    def plugin(self) -> float:
        return self._plugin

    @plugin.setter
    def plugin(self, plugin: float):
        if plugin is None:
            if hasattr(self._model, "y"):
                self._plugin = np.min(self._model.y) if self.minimize else -1.0 * np.max(self._model.y)
            else:
                self._plugin = None
        else:
            self._plugin = plugin if self.minimize else -1.0 * plugin

    def effective_plugin(self) -> float:
        if self._plugin is None:  # model had no data when the criterion was built: resolve now
            self.plugin = None
        return float(self._plugin)


class UCB(AcquisitionFunction):
    """y_hat + alpha * sd, maximised as-is (acquisition_fun.py:107-147)."""

    acq_id = _lib.ACQ_UCB

    def __init__(self, alpha: float = 0.5, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha

    @property
    def alpha(self) -> float:
        return self._alpha

    @alpha.setter
    def alpha(self, alpha: float):
        assert alpha > 0
        self._alpha = alpha

    def acq_par(self):
        return float(self._alpha)

    def __call__(self, X, return_dx: bool = False):
        X = self.check_X(X)
        f_value = self._shape(self._values(X), X.shape[0])
        if return_dx:
            _, sd = self._predict(X)
            y_dx, sd2_dx = self._gradient(X)
            with np.errstate(all="ignore"):
                sd_dx = sd2_dx / (2.0 * sd)
                f_dx = y_dx + self.alpha * sd_dx
            return f_value, f_dx
        return f_value


class EI(ImprovementBased):
    """Expected Improvement (acquisition_fun.py:150-189)."""

    acq_id = _lib.ACQ_EI

    def __call__(self, X, return_dx: bool = False):
        X = self.check_X(X)
        value = self._shape(self._values(X), X.shape[0])
        if return_dx:
            y_hat, sd = self._predict(X)
            if sd / np.sqrt(self._model.sigma2) < 1e-6:  # guard :162-164
                return 0, np.zeros((len(X[0]), 1))
            y_dx, sd2_dx = self._gradient(X)
            sd_dx = sd2_dx / (2.0 * sd)
            xcr = (self.plugin - y_hat) / sd
            dx = -y_dx * norm.cdf(xcr) + sd_dx * norm.pdf(xcr)
            return value, dx
        return value


class EpsilonPI(ImprovementBased):
    """epsilon-Probability of Improvement (acquisition_fun.py:192-228)."""

    acq_id = _lib.ACQ_EPSILON_PI
    _allow_zero = False

    def __init__(self, epsilon=1e-10, **kwargs):
        super().__init__(**kwargs)
        self.epsilon = epsilon

    @property
    def epsilon(self):
        return self._epsilon

    @epsilon.setter
    def epsilon(self, eps):
        assert eps > 0 or (self._allow_zero and eps == 0)
        self._epsilon = eps

    def acq_par(self):
        return float(self._epsilon)

    def __call__(self, X, return_dx=False):
        X = self.check_X(X)
        f_value = self._values(X).reshape(-1, 1)
        if return_dx:
            y_hat, sd = self._predict(X)
            coef = 1 - self._epsilon if y_hat > 0 else (1 + self._epsilon)
            y_dx, sd2_dx = self._gradient(X)
            with np.errstate(all="ignore"):
                sd_dx = sd2_dx / (2.0 * sd)
                xcr = (self._plugin - coef * y_hat) / sd
                f_dx = -(coef * y_dx + xcr * sd_dx) * norm.pdf(xcr) / sd
            return f_value, f_dx
        return f_value


class PI(EpsilonPI):
    """Probability of Improvement = EpsilonPI with epsilon = 0.

    The reference's PI cannot be constructed (it passes epsilon=0 into a setter that asserts eps > 0,
    acquisition_fun.py:204-206, 232-235); here epsilon = 0 is admitted for this subclass only."""

    _allow_zero = True

    def __init__(self, **kwargs):
        kwargs.update({"epsilon": 0})
        super().__init__(**kwargs)


class MGFI(ImprovementBased):
    """Moment-Generating Function of the Improvement (acquisition_fun.py:238-310)."""

    acq_id = _lib.ACQ_MGFI

    def __init__(self, t: float = 1, **kwargs):
        super().__init__(**kwargs)
        self.t = t

    @property
    def t(self) -> float:
        return self._t

    @t.setter
    def t(self, t: float):
        assert t > 0
        self._t = min(t, 22.36)  # huge t overflows (acquisition_fun.py:260-263)

    def acq_par(self):
        return float(self._t)

    def __call__(self, X, return_dx: bool = False):
        X = self.check_X(X)
        f_ = self._shape(self._values(X), X.shape[0])
        if return_dx:
            y_hat, sd = self._predict(X)
            if np.isclose(sd, 0):
                return np.array([0.0]), np.zeros((len(X[0]), 1))
            y_dx, sd2_dx = self._gradient(X)
            sd_dx = sd2_dx / (2.0 * sd)
            with np.errstate(all="raise"):
                try:
                    beta_p = (self._plugin - (y_hat - self._t * sd**2.0)) / sd
                    term = np.exp(self._t * (self._plugin + self._t * sd**2.0 / 2 - y_hat - 1))
                    m_prime_dx = y_dx - 2.0 * self._t * sd * sd_dx
                    beta_p_dx = -(m_prime_dx + beta_p * sd_dx) / sd
                    f_dx = term * (
                        norm.pdf(beta_p) * beta_p_dx + norm.cdf(beta_p) * ((self._t**2) * sd * sd_dx - self._t * y_dx)
                    )
                except Exception:
                    f_dx = np.zeros((len(X[0]), 1))
            return f_, f_dx
        return f_
