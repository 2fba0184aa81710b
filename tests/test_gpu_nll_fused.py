"""The one-launch likelihood of a small training set (csrc/kernels_nllsmall.hip: N <= 156, constant trend, one target) and the
elimination at 64-block granularity above it, up to N = 2048 (csrc/kernels_chol.hip: k_elim_*), against
(i) the general multi-kernel path of the same library (BOGP_NLL_FUSED=0) and (ii) the CPU oracle (oracle/gp_oracle.py, the
restatement of gpr.py:772-808 / :931-1038) -- every correlation family, all three estimation modes, ARD and isotropic theta,
estimated and fixed trend coefficient, sizes on every side of the 4 x 4 register blocks, a non-positive-definite matrix.
Tolerances: log-likelihood 1e-10 + 8 eps cond(R) relative (both are Cholesky-based), gradient 1e-7 + 200 eps cond(R) relative to its
largest entry.
Needs a real MI355X: `pytest -m gpu`."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bogp import _lib  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(0)
    yield e
    e.close()


def general_path(fn):
    os.environ["BOGP_NLL_FUSED"] = "0"
    try:
        return fn()
    finally:
        del os.environ["BOGP_NLL_FUSED"]


def make(N, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-3, 3, size=(N, d))
    y = np.sin(X).sum(axis=1) + 0.3 * rng.standard_normal(N)
    if N > 1:
        y = (y - y.mean()) / (y.std() + 1e-12)
    return X, y.reshape(-1, 1)


GRAD_KERNELS = [_lib.KERNEL_SE, _lib.KERNEL_MATERN12, _lib.KERNEL_MATERN32, _lib.KERNEL_MATERN52, _lib.KERNEL_ABSEXP]
SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128,   # one launch (k_nll_small, 768 threads)
         129, 131, 144, 145, 153, 156,                                            # one launch (k_nll_small, 1024 threads)
         157, 177, 192, 193, 200, 240, 249, 252,                                  # k_build_R + k_elim_* + the gradient kernels
         253, 256, 257, 320, 511, 512, 700, 1024, 1500, 1984, 2048,               # (1500: row-pair steps; 1984, 2048: the diagonal block's CU kept free)
         2049, 2500, 3072,                                                         # the elimination's far end (N <= 3072 since the end of r05)
         3073]                                                                    # the general path itself


def par_of(mode, d, iso, rng):
    if mode == _lib.MODE_NOISELESS:  # no nugget at all: short length scales keep R's condition number (and with it the distance
        return rng.uniform(2.0, 9.0, size=1 if iso else d)  # between two correct factorisations) moderate
    theta = rng.uniform(0.05, 0.6, size=1 if iso else d)
    if mode == _lib.MODE_NOISE_ESTIM:
        return np.r_[theta, 0.93]
    return np.r_[theta, 0.8]


def cond_of(kernel, mode, par, X, nv, d):
    """Condition number of the matrix both paths factorise (two correct factorisations differ by ~ eps cond)."""
    from oracle import gp_oracle

    n_theta = len(par) - (0 if mode == _lib.MODE_NOISELESS else 1)
    theta = np.full(d, par[0]) if n_theta == 1 else par[:d]
    R0 = gp_oracle.correlation_matrix(kernel, theta, X)
    if mode == _lib.MODE_NOISE_ESTIM:
        R0 = par[-1] * R0 + (1 - par[-1]) * np.eye(len(X))
    elif mode == _lib.MODE_NOISY:
        R0 = (par[-1] * R0 + nv * np.eye(len(X))) / (par[-1] + nv)
    return np.linalg.cond(R0)


def check(a, b, cond=1.0):
    la, ga = a
    lb, gb = b
    eps = 2.3e-16
    assert abs(la - lb) <= (1e-10 + 8 * eps * cond) * max(1.0, abs(lb))
    assert np.max(np.abs(ga - gb)) <= (1e-7 + 200 * eps * cond) * max(1e-300, np.max(np.abs(gb)))


@pytest.mark.parametrize("N", SIZES)
@pytest.mark.parametrize("mode", [_lib.MODE_NOISELESS, _lib.MODE_NOISE_ESTIM, _lib.MODE_NOISY])
def test_fused_equals_the_general_path(eng, N, mode):
    d = 1 + N % 7
    X, y = make(N, d, 100 + N)
    eng.set_train(X, y)
    rng = np.random.default_rng(N * 7 + mode)
    big = N > 600  # (one configuration per mode: the condition number below is an SVD of an N x N matrix)
    for kernel in ([GRAD_KERNELS[N % 5]] if big else GRAD_KERNELS):
        for iso in ((False,) if big else (False, True)):
            if iso and d == 1:
                continue
            for est in ((True,) if big else (True, False)):
                if N == 1 and est and mode == _lib.MODE_NOISELESS:
                    continue  # sigma2 = rho.rho / (N - 1): 0 / 0
                par = par_of(mode, d, iso, rng)
                nv = 0.05 if mode == _lib.MODE_NOISY else 0.0
                args = (kernel, mode, par, nv, est, 0.17)
                try:
                    want = general_path(lambda: eng.nll(*args, eval_grad=True))
                except (_lib.NotPositiveDefinite, _lib.BogpError) as e:  # noiseless + exact interpolation may be rejected (llf > 0 ...)
                    with pytest.raises(type(e)):
                        eng.nll(*args, eval_grad=True)
                    continue
                got = eng.nll(*args, eval_grad=True)
                check(got, want, cond_of(kernel, mode, par, X, nv, d))
                # value-only evaluation: the same number as with the gradient
                assert eng.nll(*args) == pytest.approx(got[0], rel=1e-13)


@pytest.mark.parametrize("kernel", [_lib.KERNEL_CUBIC, _lib.KERNEL_GENEXP])
@pytest.mark.parametrize("N", [5, 40, 128])
def test_value_only_families(eng, kernel, N):
    d = 3
    X, y = make(N, d, 7 + N)
    eng.set_train(X, y)
    theta = np.array([0.08, 0.05, 0.11])
    par = np.r_[theta, 1.7, 0.9] if kernel == _lib.KERNEL_GENEXP else np.r_[theta, 0.9]
    want = general_path(lambda: eng.nll(kernel, _lib.MODE_NOISY, par, 0.1, True, 0.0))
    got = eng.nll(kernel, _lib.MODE_NOISY, par, 0.1, True, 0.0)
    assert got == pytest.approx(want, rel=1e-10)


def test_repeated_evaluations_give_identical_bits(eng):
    X, y = make(97, 6, 5)
    eng.set_train(X, y)
    par = np.r_[np.full(6, 0.2), 0.9]
    runs = [eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True) for _ in range(5)]
    for l, g in runs[1:]:
        assert l == runs[0][0]
        np.testing.assert_array_equal(g, runs[0][1])


def test_not_positive_definite_is_reported(eng):
    """Two coincident points without a nugget: the noiseless correlation matrix is singular -- both paths refuse it."""
    X, y = make(30, 2, 9)
    X[17] = X[3]
    eng.set_train(X, y)
    par = np.array([0.3, 0.3])
    with pytest.raises(_lib.NotPositiveDefinite):
        general_path(lambda: eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISELESS, par, 0.0, True, 0.0, eval_grad=True))
    with pytest.raises(_lib.NotPositiveDefinite):
        eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISELESS, par, 0.0, True, 0.0, eval_grad=True)
    # and the handle is usable afterwards
    par = np.r_[0.3, 0.3, 0.9]
    l, g = eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISY, par, 1e-3, True, 0.0, eval_grad=True)
    assert np.isfinite(l) and np.all(np.isfinite(g))


def test_against_the_oracle(eng):
    from oracle import gp_oracle

    for N, d, seed in ((12, 2, 0), (50, 5, 1), (128, 10, 2), (150, 6, 3), (252, 12, 4), (400, 8, 5)):
        X, y = make(N, d, seed)
        eng.set_train(X, y)
        theta = np.random.default_rng(seed).uniform(0.05, 0.4, size=d)
        par = np.r_[theta, 0.85]
        l, g = eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-4, True, 0.0, eval_grad=True)
        lo, go = gp_oracle.log_likelihood_concentrated(par, X, y, _lib.KERNEL_MATERN32, _lib.MODE_NOISY, noise_var=1e-4,
                                                       estimate_trend=True, eval_grad=True)
        assert l == pytest.approx(lo, rel=1e-9)
        go = np.asarray(go).ravel()
        assert np.max(np.abs(g - go)) <= 1e-6 * np.max(np.abs(go))


def test_large_d_falls_back_when_the_workgroup_would_not_fit(eng):
    """N = 150, d = 40: X + the image of the blocks exceed the LDS of one CU -- the evaluation takes the general path, same numbers."""
    X, y = make(150, 40, 11)
    eng.set_train(X, y)
    par = np.r_[np.full(40, 0.02), 0.9]
    got = eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-4, True, 0.0, eval_grad=True)
    want = general_path(lambda: eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-4, True, 0.0, eval_grad=True))
    check(got, want, 1e3)


def test_extremes_of_theta_and_of_the_dimension(eng):
    """d = 64 (the kernel-argument limit), theta so small that R is nearly rank one and so large that it is nearly the identity."""
    X, y = make(60, 64, 21)
    eng.set_train(X, y)
    for th in (1e-4, 0.3, 50.0):
        par = np.r_[np.full(64, th / 64), 0.7]
        args = (_lib.KERNEL_SE, _lib.MODE_NOISY, par, 1e-3, True, 0.0)
        got = eng.nll(*args, eval_grad=True)
        want = general_path(lambda: eng.nll(*args, eval_grad=True))
        check(got, want, cond_of(_lib.KERNEL_SE, _lib.MODE_NOISY, par, X, 1e-3, 64))
    X, y = make(90, 1, 22)
    eng.set_train(X, y)
    for th in (1e-3, 1.0, 1e3):
        par = np.r_[th, 0.9]
        args = (_lib.KERNEL_MATERN52, _lib.MODE_NOISE_ESTIM, par, 0.0, False, -0.3)
        got = eng.nll(*args, eval_grad=True)
        want = general_path(lambda: eng.nll(*args, eval_grad=True))
        check(got, want, cond_of(_lib.KERNEL_MATERN52, _lib.MODE_NOISE_ESTIM, par, X, 0.0, 1))


def test_duplicate_points_with_a_nugget_and_positive_likelihood_rejection(eng):
    """Coincident rows are fine with noise on the diagonal; a likelihood > 0 is rejected exactly where the general path rejects it."""
    X, y = make(40, 3, 23)
    X[7] = X[8] = X[9]
    eng.set_train(X, y)
    par = np.r_[0.2, 0.2, 0.2, 0.8]
    args = (_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 0.05, True, 0.0)
    check(eng.nll(*args, eval_grad=True), general_path(lambda: eng.nll(*args, eval_grad=True)), 1e3)
    # tiny targets: sigma2_total small -> llf > 0 (gpr.py:981-982)
    eng.set_train(X, 1e-4 * y)
    par = np.r_[0.2, 0.2, 0.2]
    for fn in (lambda: eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISELESS, np.r_[5.0, 5.0, 5.0], 0.0, True, 0.0, eval_grad=True),):
        try:
            want = general_path(fn)
            got = fn()
            check(got, want, 1e3)
        except _lib.BogpError as e:
            with pytest.raises(type(e)):
                fn()


def test_many_evaluations_on_two_engines_interleaved():
    """Each engine has its own pinned read-back block and sequence counter: interleaved calls do not see each other's results."""
    e1, e2 = _lib.Engine(0), _lib.Engine(0)
    try:
        X1, y1 = make(33, 4, 31)
        X2, y2 = make(120, 7, 32)
        e1.set_train(X1, y1)
        e2.set_train(X2, y2)
        p1, p2 = np.r_[np.full(4, 0.2), 0.9], np.r_[np.full(7, 0.1), 0.8]
        r1 = e1.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, p1, 1e-4, True, 0.0, eval_grad=True)
        r2 = e2.nll(_lib.KERNEL_SE, _lib.MODE_NOISE_ESTIM, p2, 0.0, True, 0.0, eval_grad=True)
        for _ in range(300):
            a = e1.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, p1, 1e-4, True, 0.0, eval_grad=True)
            b = e2.nll(_lib.KERNEL_SE, _lib.MODE_NOISE_ESTIM, p2, 0.0, True, 0.0, eval_grad=True)
            assert a[0] == r1[0] and b[0] == r2[0]
            np.testing.assert_array_equal(a[1], r1[1])
            np.testing.assert_array_equal(b[1], r2[1])
    finally:
        e1.close()
        e2.close()


def test_commit_after_a_fused_evaluation(eng):
    """bogp_commit runs the general path (it must leave the factor buffers): its likelihood equals the fused one."""
    X, y = make(80, 4, 3)
    eng.set_train(X, y)
    par = np.r_[np.full(4, 0.15), 0.9]
    l = eng.nll(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-5, True, 0.0)
    lc = eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, par, 1e-5, True, 0.0)
    assert l == pytest.approx(lc, rel=1e-11)
