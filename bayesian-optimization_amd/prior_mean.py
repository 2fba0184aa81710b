"""Prior-mean (trend) bases for the GPU GaussianProcess.

API mirror of the reference's trend objects (`bayes_optim/surrogate/gaussian_process/trend.py:10-142`): the same
names (`constant_trend`, `linear_trend`, `quadratic_trend`), constructor arguments and `beta` convention --
`beta=None` asks for the GLS estimate (ordinary / universal kriging), a number fixes it (simple kriging,
`gpr.py:269-270`).  Implementation differs: one table-driven class instead of a hierarchy.

Device support: all three bases are evaluated by libbogp (`BOGP_TREND_CONSTANT / LINEAR / QUADRATIC`); the constant
basis keeps a scalar fast path, p > 1 goes through a CholeskyQR2 of `L^-1 F` on the device.
"""
from __future__ import annotations

import numpy as np


def _basis_constant(X):
    return np.ones((X.shape[0], 1))


def _basis_linear(X):
    return np.hstack([np.ones((X.shape[0], 1)), X])


def _basis_quadratic(X):
    cols = [np.ones((X.shape[0], 1)), X]
    cols += [X[:, k : k + 1] * X[:, k:] for k in range(X.shape[1])]  # all x_i x_j with j >= i (trend.py:130-136)
    return np.hstack(cols)


_BASES = {
    "constant": (_basis_constant, lambda n: 1),
    "linear": (_basis_linear, lambda n: n + 1),
    "quadratic": (_basis_quadratic, lambda n: (n + 1) * (n + 2) // 2),
}


class Trend:
    """F(X) beta with a polynomial basis F of `n_dim` columns over `n_feature` inputs."""

    kind = "constant"

    def __init__(self, n_feature, beta=None):
        self.n_feature = int(n_feature)
        self._F, size = _BASES[self.kind]
        self.n_dim = int(size(self.n_feature))
        self.beta = beta

    # beta is stored as a (p, 1) column or None, like the reference's property (trend.py:17-29)
    @property
    def beta(self):
        return self._beta

    @beta.setter
    def beta(self, value):
        if value is not None:
            value = np.full(self.n_dim, value, dtype=float) if np.ndim(value) == 0 else np.asarray(value, dtype=float)
            value = value.reshape(-1, 1)
            if value.shape[0] != self.n_dim:
                raise Exception("Shapes of beta and F do not match.")
        self._beta = value

    def check_input(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        if X.shape[1] != self.n_feature:
            X = X.T
        if X.shape[1] != self.n_feature:
            raise Exception("X does not have the right size!")
        return X

    def F(self, X):
        return self._F(self.check_input(X))

    def __call__(self, X):
        if self._beta is None:
            raise Exception("beta is not set!")
        return self.F(X) @ self._beta

    def Jacobian(self, x):
        x = self.check_input(x)
        if self.kind == "constant":
            return np.zeros((1, self.n_feature))
        if self.kind == "linear":
            return np.vstack([np.zeros((1, self.n_feature)), np.eye(self.n_feature)])
        raise NotImplementedError  # as the reference (trend.py:138-139)

    def __getstate__(self):
        st = dict(self.__dict__)
        st.pop("_F", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._F = _BASES[self.kind][0]


class constant_trend(Trend):
    kind = "constant"


class linear_trend(Trend):
    kind = "linear"


class quadratic_trend(Trend):
    kind = "quadratic"


_TREND_IDS = {"constant_trend": 0, "linear_trend": 1, "quadratic_trend": 2}  # BOGP_TREND_* of include/bogp.h


def device_trend_of(mean):
    """Map a trend object (ours or the reference's, duck-typed by class name) to libbogp's arguments:
    returns (trend_id, estimate_trend: bool, beta) with beta a float for the constant basis and a (p,) vector
    otherwise.  Raises NotImplementedError for bases the device does not evaluate (e.g. NonparametricTrend)."""
    name = type(mean).__name__
    if name not in _TREND_IDS:
        raise NotImplementedError("trend %r is not built on the device (constant, linear and quadratic bases are)" % name)
    tid = _TREND_IDS[name]
    b = mean.beta
    if b is None:
        return tid, True, 0.0
    b = np.asarray(b, dtype=float).ravel()
    return tid, False, (float(b[0]) if tid == 0 else b)
