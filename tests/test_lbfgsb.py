"""The optimiser of the lock-step MLE (csrc/bogp_lbfgsb.h: L-BFGS-B after Byrd, Lu, Nocedal & Zhu 1995 + the More'-Thuente line
search, written as a re-entrant state machine) against scipy.optimize.fmin_l_bfgs_b -- what the reference's restart loop calls
(gpr.py:1136) -- through bogp_lbfgsb_minimize, on the CPU: same published algorithm and defaults, so from the same start it must
reach the same minimiser with a comparable number of evaluations.  No GPU needed."""
import numpy as np
import pytest
from scipy.optimize import fmin_l_bfgs_b

from bogp import _lib


def rosen(x):
    x = np.asarray(x)
    f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] += -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


@pytest.mark.parametrize("n", [2, 5, 10, 20])
@pytest.mark.parametrize("box", [(-2.0, 2.0), (-1.5, 0.8)])  # the second box cuts the unconstrained minimiser (1, ..., 1) off
def test_rosenbrock_in_a_box_like_scipy(n, box):
    rng = np.random.default_rng(n * 3 + int(box[1] * 10))
    b = [box] * n
    for _ in range(3):
        x0 = rng.uniform(box[0], box[1], n)
        xs, fs, ds = fmin_l_bfgs_b(rosen, x0, bounds=b)
        xo, fo, do = _lib.lbfgsb_minimize(rosen, x0, b)
        assert do["status"] in (0, 1)
        assert np.all(xo >= box[0]) and np.all(xo <= box[1])
        assert fo <= fs + 1e-7 * max(1.0, abs(fs))
        assert np.max(np.abs(xo - xs)) < 1e-3
        assert do["funcalls"] <= 1.25 * ds["funcalls"] + 5


@pytest.mark.parametrize("n", [3, 11, 30])
def test_convex_quadratic_with_active_bounds_like_scipy(n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    A = A @ A.T + 0.1 * np.eye(n)
    c = 3 * rng.standard_normal(n)
    fun = lambda x: (0.5 * x @ A @ x - c @ x, A @ x - c)  # noqa: E731
    b = [(-0.5, 0.7)] * n
    x0 = rng.uniform(-0.5, 0.7, n)
    xs, fs, ds = fmin_l_bfgs_b(fun, x0, bounds=b, factr=10, pgtol=1e-10)
    xo, fo, do = _lib.lbfgsb_minimize(fun, x0, b, factr=10, pgtol=1e-10)
    assert abs(fo - fs) <= 1e-10 * max(1.0, abs(fs))
    np.testing.assert_allclose(xo, xs, atol=1e-7)
    assert (np.isclose(xo, -0.5) | np.isclose(xo, 0.7)).any()  # (the test is about ACTIVE bounds)
    # first-order optimality of the bound-constrained problem: the projected gradient vanishes
    g = A @ xo - c
    pg = np.where(g < 0, np.maximum(xo - 0.7, g), np.minimum(xo + 0.5, g))
    assert np.max(np.abs(pg)) < 1e-6
    assert do["funcalls"] <= 1.25 * ds["funcalls"] + 5


def test_start_outside_the_box_is_clipped_and_a_start_at_the_solution_stops_at_once():
    fun = lambda x: (float(np.sum((x - 3.0) ** 2)), 2 * (x - 3.0))  # noqa: E731
    b = [(0.0, 1.0)] * 4
    xo, fo, do = _lib.lbfgsb_minimize(fun, np.full(4, 9.0), b)
    np.testing.assert_array_equal(xo, np.ones(4))  # the minimiser of the box: the upper corner
    assert do["funcalls"] == 1 and do["status"] == 0  # projected gradient zero at the clipped start


def test_infinite_objective_regions_are_stepped_back_from():
    """The MLE's objective is +inf (zero gradient) where the factorisation breaks down (gpr.py:946-947): the line search must
    shorten the step instead of accepting or diverging."""
    def fun(x):
        if x[0] > 1.5:  # a wall right behind the minimiser's neighbourhood
            return np.inf, np.zeros_like(x)
        return float(np.sum((x - 1.2) ** 2) + 0.1 * np.sum(x**4)), 2 * (x - 1.2) + 0.4 * x**3

    b = [(-4.0, 4.0)] * 3
    xo, fo, do = _lib.lbfgsb_minimize(fun, np.array([-3.5, 3.0, -2.0]), b)
    assert np.isfinite(fo) and do["status"] in (0, 1)
    xs, fs, _ = fmin_l_bfgs_b(lambda x: (float(np.sum((x - 1.2) ** 2) + 0.1 * np.sum(x**4)), 2 * (x - 1.2) + 0.4 * x**3), np.zeros(3), bounds=b)
    assert abs(fo - fs) < 1e-6


def test_start_in_an_infinite_region_is_reported():
    fun = lambda x: (np.inf, np.zeros_like(x))  # noqa: E731
    xo, fo, do = _lib.lbfgsb_minimize(fun, np.zeros(2), [(-1, 1)] * 2)
    assert do["status"] == 5 and np.isinf(fo) and do["funcalls"] == 1


def test_budget_is_tested_at_iterates_like_scipy():
    """maxfun: scipy stops at the first ACCEPTED iterate after the count exceeds it (never inside a line search), and returns
    that iterate; so do we -- same count, same point."""
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-2, 2, 10)
    b = [(-2.0, 2.0)] * 10
    for maxfun in (5, 12, 30):
        xs, fs, ds = fmin_l_bfgs_b(rosen, x0, bounds=b, maxfun=maxfun)
        xo, fo, do = _lib.lbfgsb_minimize(rosen, x0, b, maxfun=maxfun)
        assert do["status"] == 2 and ds["warnflag"] == 1
        assert abs(do["funcalls"] - ds["funcalls"]) <= 2
        assert fo <= rosen(x0)[0]
        assert abs(fo - fs) <= 1e-6 * max(1.0, abs(fs)) or do["funcalls"] != ds["funcalls"]


def test_invalid_arguments_are_refused():
    with pytest.raises(_lib.BogpError):
        _lib.lbfgsb_minimize(rosen, np.zeros(3), [(1.0, -1.0)] * 3)


@pytest.mark.parametrize("bounds", ["boxed", "open", "lower only"])
@pytest.mark.parametrize("shape", ["quadratic", "quartic"])
def test_evaluation_points_follow_scipy_point_by_point(bounds, shape):
    """MINPACK-2's dcsrch hands dcstep the MOVING interval [stmin, stmax]: an un-bracketed extrapolation is capped at
    stp + 4 (stp - stx), not at the global stpmax; and lnsrlb's first-step rules depend on the bounds (unconstrained: first trial 1 / |d|,
    stpmax 1e10; constrained: stpmax = 1 on the first iteration; every variable boxed: first trial 1).  A shallow bowl far from its
    minimiser makes the searches extrapolate: the SEQUENCE of evaluation points must be scipy's, point by point, in all three regimes."""
    n = 4
    c = np.array([30.0, -45.0, 60.0, 20.0])
    fn = {"quadratic": lambda x: (float(0.001 * np.sum((x - c) ** 2)), 0.002 * (x - c)),
          "quartic": lambda x: (float(1e-6 * np.sum((x - c) ** 4)), 4e-6 * (x - c) ** 3)}[shape]  # fmt: skip

    def make():
        seen = []

        def fun(x):
            seen.append(np.array(x, dtype=float))
            return fn(x)

        return fun, seen

    b = {"boxed": [(-500.0, 500.0)] * n, "open": [(None, None)] * n, "lower only": [(-500.0, None)] * n}[bounds]
    f1, s1 = make()
    xs, fs, ds = fmin_l_bfgs_b(f1, np.zeros(n), bounds=b)
    f2, s2 = make()
    xo, fo, do = _lib.lbfgsb_minimize(f2, np.zeros(n), [(-1e300 if lo is None else lo, 1e300 if hi is None else hi) for lo, hi in b])
    assert len(s2) == len(s1) == ds["funcalls"] == do["funcalls"]
    for i in range(len(s1)):
        np.testing.assert_allclose(s2[i], s1[i], rtol=1e-7, atol=1e-9, err_msg="evaluation %d" % i)
    if bounds == "open":  # the first search extrapolates by the capped factor: |x_2| = 5 |x_1| (stmax = stp + 4 stp), not to stpmax = 1e10
        r = [np.linalg.norm(v) for v in s2[1:3]]
        np.testing.assert_allclose([r[1] / r[0]], [5.0], rtol=1e-12)
    np.testing.assert_allclose(xo, xs, rtol=1e-7, atol=1e-7)
