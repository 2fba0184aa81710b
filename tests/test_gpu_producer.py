"""The correlation producer with the distance on the matrix cores (k_corr_mfma, csrc/kernels_posterior.hip; r05): the weighted squared
distance as |a|^2 + |b|^2 - 2 a.b for far pairs, in the reference's difference form (gpr.py:42-47 -> kernel.py:186-200, 289-329) for
pairs closer than an eighth of their norm scale.  These tests aim at the seam:
  * candidates ON training points, a hair beside them (1e-12 .. 1e-2 relative) and far away, in one sweep, N > 512 so that the chunked
    path runs: posterior against the oracle at the parity tolerances, exact zeros of the clipped MSE where the reference has them;
  * un-centred data (every coordinate near +100: |a|^2 + |b|^2 is 1e4 times any distance, EVERY pair takes the difference form);
  * d not a multiple of four (zero-padded k-steps), d = 1, N not a multiple of 64 (idle waves in the last block).
and at the other squared-distance kernels through the same code."""
import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

from bogp import _lib  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(0)
    yield e
    e.close()


def _check(eng, X, y, par, kernel, Xs, nv=1e-6, est=False):
    eng.set_train(X, y)
    eng.commit(kernel, O.MODE_NOISY, par, nv, est, 0.0)
    eng.upload_candidates(Xs)
    mu, mse = eng.predict()
    st = O.make_state(par, X, y, kernel, O.MODE_NOISY, nv, estimate_trend=est)
    omu, omse = O.predict_chunked(st, Xs, 512)
    s2 = float(st.sigma2[0])
    np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * s2)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0), (_lib.ACQ_UCB, 0.5)]
    pl = float(y.min())
    best, idx = eng.sweep(acq, pl, True)
    obest, oidx = O.sweep(st, Xs, acq, pl, True)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_allclose(best, obest, rtol=1e-6)
    return mu, mse, omu.ravel(), omse.ravel()


@pytest.mark.parametrize("kernel", [O.KERNEL_SE, O.KERNEL_MATERN12, O.KERNEL_MATERN32, O.KERNEL_MATERN52])
@pytest.mark.parametrize("N,d", [(600, 20), (1000, 7), (777, 1), (530, 50)])
def test_near_and_far_pairs_in_one_sweep(eng, kernel, N, d):
    rng = np.random.default_rng(1000 * kernel + N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1) + 0.05 * rng.standard_normal((N, 1))  # (noise keeps llf <= 0: gpr.py rejects positive values)
    par = np.r_[np.full(d, 0.2 / d if d > 1 else 500.0), 0.9]  # (d = 1: 777 points on a line need a short length scale to stay positive definite)
    parts = [X[:200].copy()]  # ON training points
    for rel in (1e-12, 1e-9, 1e-6, 1e-4, 1e-3, 1e-2, 1e-1):
        parts.append(X[200:320] * (1.0 + rel * rng.standard_normal((120, d))))
    parts.append(rng.uniform(-5, 5, size=(3000, d)))
    parts.append(rng.uniform(-60, 60, size=(200, d)))  # far: r underflows towards 0
    Xs = np.vstack(parts)
    mu, mse, omu, omse = _check(eng, X, y, par, kernel, Xs)
    # a candidate on a training point: the difference form gives s2 = 0 exactly, r = 1, and the MSE is the nugget's share only
    assert np.all(mse[:200] < 1e-5) and np.all(mse[:200] >= 0.0)
    np.testing.assert_allclose(mse[:200], omse[:200], rtol=1e-6, atol=1e-12)


def test_noiseless_duplicates_give_exactly_zero_variance(eng):
    """No nugget (noiseless mode): at a training point the reference's MSE is clipped to exactly 0 and EI / MGFI return exact zeros
    (acquisition_fun.py:162-164, 274-275) -- only if r(x_i, x_i) is exactly 1, i.e. s2 exactly 0."""
    rng = np.random.default_rng(5)
    N, d = 640, 3
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sin(X).sum(axis=1).reshape(-1, 1)
    par = np.full(d, 2.0)
    eng.set_train(X, y)
    eng.commit(O.KERNEL_MATERN32, O.MODE_NOISELESS, par, 0.0, False, 0.0)
    Xs = np.vstack([X[:256], rng.uniform(-5, 5, (1000, d))])
    eng.upload_candidates(Xs)
    mu, mse = eng.predict()
    st = O.make_state(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISELESS, 0.0)
    omu, omse = O.predict_chunked(st, Xs, 512)
    assert np.count_nonzero(omse[:256]) < 256  # the reference clips some of them to exactly 0
    np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))
    _, _, vals = eng.sweep([(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0)], float(y.min()), True, return_values=True)
    ovals = np.array([O.acquisition(a, p, omu.ravel(), omse.ravel(), float(y.min()), float(st.sigma2[0]), True) for a, p in ((O.ACQ_EI, 0.0), (O.ACQ_MGFI, 2.0))])
    # EI's guard (sd / sqrt(sigma2) < 1e-6) fires on every duplicate on both sides; MGFI's (|sd| <= 1e-8) sits AT the rounding noise of
    # 1 - sum rt^2 there (1e-16 sigma2 with either sign), so neither its zero pattern nor its values are comparable on those rows
    assert np.all(vals[0, :256] == 0.0) and np.all(ovals[0, :256] == 0.0)
    assert np.all(np.isfinite(vals[1, :256])) and np.all(vals[1, :256] >= 0.0)  # (either side of a guard that is a step function of the noise)
    np.testing.assert_allclose(vals[:, 256:], ovals[:, 256:], rtol=1e-6, atol=1e-300)


def test_uncentred_data_takes_the_difference_form_everywhere(eng):
    rng = np.random.default_rng(8)
    N, d = 700, 6
    X = 100.0 + rng.uniform(-1, 1, size=(N, d))
    y = np.sum((X - 100.0) ** 2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 1.5), 0.9]
    Xs = 100.0 + rng.uniform(-1, 1, size=(5000, d))
    _check(eng, X, y, par, O.KERNEL_MATERN52, Xs)
    _check(eng, X, y, par, O.KERNEL_SE, Xs, est=True)


@pytest.mark.parametrize("N,d,tid", [(700, 8, O.TREND_QUADRATIC), (768, 7, O.TREND_QUADRATIC), (300, 40, O.TREND_LINEAR), (2048, 9, O.TREND_QUADRATIC)])
def test_wide_polynomial_bases_ride_inside_the_contraction(eng, N, d, tid):
    """Universal kriging with more than 32 basis columns (r05, kernels_fit.hip: k_pack_Vx): the u term of gpr.py:496-498 as p extra rows of
    the packed triangular factor, the chunk extended by -f(x*).  Posterior, criteria and argmax against the oracle at the parity
    tolerances: N off and on the 256-row group boundary (the zero 'hole' rows), a wide LINEAR basis (d = 40: p = 41), the C3 row count."""
    rng = np.random.default_rng(N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + X[:, 0] - 0.3 * X[:, 1] * X[:, 2]
    y = ((y - y.mean()) / y.std()).reshape(-1, 1) + 0.05 * rng.standard_normal((N, 1))
    par = np.r_[np.full(d, 0.3 / d), 0.9]
    Xs = np.vstack([rng.uniform(-5, 5, size=(4000, d)), X[:50], rng.uniform(-9, 9, size=(200, d))])
    p = _lib.trend_size_of(tid, d)
    assert p > 32
    eng.set_train(X, y)
    eng.commit(O.KERNEL_MATERN52, O.MODE_NOISY, par, 1e-6, True, 0.0, trend=tid)
    eng.upload_candidates(Xs)
    mu, mse = eng.predict()
    st = O.make_state(par, X, y, O.KERNEL_MATERN52, O.MODE_NOISY, 1e-6, trend=tid, estimate_trend=True)
    omu, omse = O.predict_chunked(st, Xs, 512)
    s2 = float(st.sigma2[0])
    np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * s2)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_UCB, 0.5)]
    best, idx = eng.sweep(acq, float(y.min()), True)
    obest, oidx = O.sweep(st, Xs, acq, float(y.min()), True)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_allclose(best, obest, rtol=1e-6)
    assert eng.last_timing()["acquisition_ms"] < 5.0  # (the tile products of the old path showed up here)
