"""Parity of the one-point / B-point consumption path (csrc/kernels_point.hip, r03): `bogp_point_eval_batch` -- posterior,
input-gradients, criteria AND the criteria's own input-gradients for B rows in one device round trip -- against the
reference's goldens (`gradient()`, `criterion(x, return_dx=True)` at 8 stored points: gpr.py:537-576,
acquisition_fun.py:139-146, 181-188, 220-227, 292-309) at north_star's 1e-6; and `bogp_polish`, the lock-step multi-start
refinement, against the reference-style sequential L-BFGS-B from the same starts.  Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
from scipy.optimize import fmin_l_bfgs_b

from conftest import load_golden, state_from_golden
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

import bogp  # noqa: E402
from bogp import _lib  # noqa: E402

DX_FILES = ["G1_se_sk_noisy", "G2_m32_ok_noisy", "G4_se_ok_noiseless", "G5_se_sk_noise_estim", "G12_absexp_ok_noisy"]
DX_ACQ = [("EI", O.ACQ_EI, 0.0), ("EpsilonPI", O.ACQ_EPSILON_PI, 1e-10), ("UCB", O.ACQ_UCB, 0.5), ("MGFI_2", O.ACQ_MGFI, 2.0)]


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(0)
    yield e
    e.close()


def commit_golden(eng, g):
    mode, kernel = int(g["mode"]), int(g["kernel"])
    nv = float(g["noise_var"][0]) if mode == O.MODE_NOISY else 0.0
    eng.set_train(g["X"], g["y"])
    return eng.commit(kernel, mode, g["par"], nv, bool(g["estimate_trend"]), 0.0)


@pytest.mark.parametrize("name", DX_FILES)
def test_batched_return_dx_matches_the_reference(eng, name):
    """batched gradients vs the reference's one-row `return_dx` outputs, 1e-6."""
    g = load_golden(name)
    if "dx_EI" not in g:
        pytest.skip("fixture holds no return_dx rows")
    commit_golden(eng, g)
    nb = len(g["grad_mu"])  # 8 stored points (4 in the noiseless fixture)
    Xb = g["Xs"][:nb]
    acq = [(i, p) for _, i, p in DX_ACQ]
    mu, mse, dmu, dmse, vals, dvals = eng.point_eval_batch(Xb, acq, float(g["plugin_eff"][0]), True)
    np.testing.assert_allclose(mu, g["mu"][:nb, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, g["mse"][:nb, 0], rtol=1e-6, atol=1e-12 * float(g["sigma2"][0]))
    np.testing.assert_allclose(dmu, g["grad_mu"][:, :, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(dmse, g["grad_mse"][:, :, 0], rtol=1e-6, atol=1e-9 * float(g["sigma2"][0]))
    for c, (key, _, _) in enumerate(DX_ACQ):
        np.testing.assert_allclose(vals[:, c], g["dx_val_" + key], rtol=1e-6, atol=1e-300, err_msg=key)
        np.testing.assert_allclose(dvals[:, c, :], g["dx_" + key], rtol=1e-6, atol=1e-12, err_msg=key)
    # the same rows one at a time, through the one-point entry (kernel-argument path, B = 1) and the plain gradients
    for i in range(nb):
        m1, s1, a1, b1, v1 = eng.point_eval(Xb[i], acq, float(g["plugin_eff"][0]), True)
        # same kernels; one point is evaluated in latency mode (row blocks split over several workgroups), so the order of
        # the additions -- not the result beyond rounding -- differs from the batch's
        np.testing.assert_allclose([m1, s1], [mu[i], mse[i]], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(a1, dmu[i], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(b1, dmse[i], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(v1, vals[i], rtol=1e-9, atol=1e-300)
        m2, s2, a2, b2, v2 = eng.point_eval(Xb[i], acq, float(g["plugin_eff"][0]), True)
        assert (m1, s1) == (m2, s2) and np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(v1, v2)  # deterministic
    gb = eng.gradient_batch(Xb)
    np.testing.assert_array_equal(gb[0], dmu)
    np.testing.assert_array_equal(gb[1], dmse)


@pytest.mark.parametrize("N,d,kernel,est", [(300, 3, O.KERNEL_SE, False), (700, 25, O.KERNEL_MATERN32, True), (1100, 11, O.KERNEL_ABSEXP, True),
                                            (2048, 20, O.KERNEL_MATERN12, False), (97, 47, O.KERNEL_SE, True)])  # fmt: skip
def test_point_batch_against_the_oracle_at_other_shapes(eng, N, d, kernel, est):
    """Shapes the goldens do not have: several passes over the right-hand sides (d > 21), ragged 16-row blocks (N not a
    multiple of 16 / 32 / 64), the 12- and 22-column instantiations, ordinary kriging.  Maximising orientation included."""
    rng = np.random.default_rng(N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.6 / d), 0.8]
    eng.set_train(X, y)
    eng.commit(kernel, O.MODE_NOISY, par, 1e-4, est, 0.0)
    st = O.make_state(par, X, y, kernel, O.MODE_NOISY, 1e-4, estimate_trend=est, beta=None if est else 0.0)
    Xb = np.vstack([rng.uniform(-5, 5, size=(5, d)), X[:2] + 1e-2 * rng.standard_normal((2, d))])
    for minimize in (True, False):
        pl = O.plugin_value(st.y, minimize)
        acq = [(O.ACQ_EI, 0.0), (O.ACQ_UCB, 0.7), (O.ACQ_MGFI, 1.5), (O.ACQ_EPSILON_PI, 1e-3)]
        mu, mse, dmu, dmse, vals, dvals = eng.point_eval_batch(Xb, acq, pl, minimize)
        omu, omse = O.predict(st, Xb)
        np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))
        gp = bogp.GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d)  # host chain rule, reference's form
        for i in range(len(Xb)):
            odmu, odmse = O.gradient(st, Xb[i])
            np.testing.assert_allclose(dmu[i], np.ravel(odmu), rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(dmse[i], np.ravel(odmse), rtol=1e-6, atol=1e-9 * float(st.sigma2[0]))
            # the acquisition chain rule of the reference on the ORACLE's moments (host, acquisition._dx) vs the device's
            for c, (aid, apar) in enumerate(acq):
                cls = {O.ACQ_EI: bogp.EI, O.ACQ_UCB: bogp.UCB, O.ACQ_MGFI: bogp.MGFI, O.ACQ_EPSILON_PI: bogp.EpsilonPI}[aid]
                kw = {O.ACQ_UCB: {"alpha": apar}, O.ACQ_MGFI: {"t": apar}, O.ACQ_EPSILON_PI: {"epsilon": apar}}.get(aid, {})
                gp.sigma2, gp.y = st.sigma2, st.y
                crit = cls(model=gp, minimize=minimize, **kw)
                mom = bogp.acquisition._Moments(crit, Xb[i : i + 1], (omu[i : i + 1], omse[i : i + 1], np.reshape(odmu, (-1, 1)), np.reshape(odmse, (-1, 1))))
                ov = O.acquisition(aid, apar, omu[i : i + 1], omse[i : i + 1], pl, float(st.sigma2[0]), minimize)
                _, odx = crit._dx(mom, np.ravel(ov))
                scale = max(1e-300, float(np.max(np.abs(odx))))
                np.testing.assert_allclose(vals[i, c], np.ravel(ov)[0], rtol=1e-6, atol=1e-300)
                np.testing.assert_allclose(dvals[i, c], np.ravel(odx), rtol=2e-6, atol=1e-9 * scale, err_msg="acq %d row %d" % (aid, i))


@pytest.mark.parametrize("N,d,kernel,est,B", [(300, 3, O.KERNEL_SE, False, 37), (700, 25, O.KERNEL_MATERN32, True, 9), (1100, 11, O.KERNEL_ABSEXP, True, 64),
                                              (2048, 20, O.KERNEL_MATERN52, False, 130), (97, 47, O.KERNEL_SE, True, 3), (513, 63, O.KERNEL_MATERN12, True, 5),
                                              (260, 15, O.KERNEL_SE, False, 4), (260, 16, O.KERNEL_SE, True, 4), (64, 31, O.KERNEL_MATERN32, False, 2)])  # fmt: skip
def test_mfma_flavour_of_the_point_batch_equals_the_valu_flavour_and_the_oracle(eng, N, d, kernel, est, B):
    """r03: with enough right-hand sides the B-point path runs C = V rhs through k_contract16's cross-product epilogue (FP64 MFMA
    against the packed V; 16 / 32 / 64 columns per point) instead of k_point_tri.  Both flavours on the same points
    (BOGP_POINT_MFMA_MIN = 1 / 0 forces one or the other): the same numbers up to the order of the sums, the oracle's at 1e-6 --
    shapes on both sides of every column count (d + 1 = 16, 17, 32, 64), ragged N, one and many 64-column tiles, padding points."""
    import os

    rng = np.random.default_rng(31 * N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(np.sin(X), axis=1) + 0.1 * rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.5 / d), 0.85]
    eng.set_train(X, y)
    eng.commit(kernel, O.MODE_NOISY, par, 1e-5, est, 0.0)
    st = O.make_state(par, X, y, kernel, O.MODE_NOISY, 1e-5, estimate_trend=est, beta=None if est else 0.0)
    Xb = rng.uniform(-5, 5, size=(B, d))
    Xb[B // 2] = X[1] + 1e-3 * rng.standard_normal(d)
    pl = O.plugin_value(st.y, True)
    acq = [(O.ACQ_EI, 0.0), (O.ACQ_MGFI, 2.0), (O.ACQ_UCB, 0.5)]
    out = {}
    for tag, v in (("mfma", "1"), ("valu", "0")):
        os.environ["BOGP_POINT_MFMA_MIN"] = v
        try:
            out[tag] = eng.point_eval_batch(Xb, acq, pl, True)
            again = eng.point_eval_batch(Xb, acq, pl, True)
        finally:
            del os.environ["BOGP_POINT_MFMA_MIN"]
        for a, b in zip(out[tag], again):
            np.testing.assert_array_equal(a, b)  # deterministic
    s2 = float(st.sigma2[0])
    names = ("mu", "mse", "dmu", "dmse", "values", "dvalues")
    for name, a, b in zip(names, out["mfma"], out["valu"]):
        scale = max(1e-300, float(np.max(np.abs(b))))
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-9 * scale, err_msg=name)  # two summation orders of the same sums
    omu, omse = O.predict(st, Xb)
    np.testing.assert_allclose(out["mfma"][0], omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out["mfma"][1], omse.ravel(), rtol=1e-6, atol=1e-12 * s2)
    for i in range(min(B, 4)):
        odmu, odmse = O.gradient(st, Xb[i])
        np.testing.assert_allclose(out["mfma"][2][i], np.ravel(odmu), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(out["mfma"][3][i], np.ravel(odmse), rtol=1e-6, atol=1e-9 * s2)


def test_acquisition_call_with_many_rows_and_return_dx(eng):
    """`criterion(X, return_dx=True)` with several rows (the reference raises, gpr.py:548-549): row i = its one-row answer."""
    g = load_golden("G1_se_sk_noisy")
    d = g["X"].shape[1]
    gp = bogp.GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    gp.set_state(g["par"], g["X"], g["y"])
    for cls, key, kw in ((bogp.EI, "EI", {}), (bogp.MGFI, "MGFI_2", {"t": 2}), (bogp.UCB, "UCB", {}), (bogp.EpsilonPI, "EpsilonPI", {})):
        c = cls(model=gp, minimize=True, **kw)
        v, dx = c(g["Xs"][:8], return_dx=True)
        assert v.shape == (8, 1) and dx.shape == (8, d)
        np.testing.assert_allclose(v.ravel(), g["dx_val_" + key], rtol=1e-6)
        np.testing.assert_allclose(dx, g["dx_" + key], rtol=1e-6, atol=1e-12)


def _c3_like(eng, N=2048, d=20, kernel=O.KERNEL_MATERN52):
    rng = np.random.default_rng(0)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.01), 0.9]
    eng.set_train(X, y)
    eng.commit(kernel, O.MODE_NOISY, par, 1e-6, False, 0.0)
    return X, y, rng


@pytest.mark.parametrize("acq", [(O.ACQ_EI, 0.0), (O.ACQ_MGFI, 2.0), (O.ACQ_UCB, 0.5)])
def test_polish_is_monotone_and_at_least_as_good_as_sequential_lbfgsb(eng, acq):
    """VERDICT r02 item 3 "done" test: the lock-step device polish of the sweep's top-32 at C3 size ends at least as high
    as the reference-style loop (scipy L-BFGS-B, pgtol 1e-8, factr 1e6, maxfun 50 per start: optim/__init__.py:94-101)
    run one start after the other on the same starts through the one-point device call."""
    X, y, rng = _c3_like(eng)
    d = X.shape[1]
    pl = float(y.min())
    Xs = rng.uniform(-5, 5, size=(200_000, d))
    eng.upload_candidates(Xs)
    tv, ti = eng.sweep_topk([acq], pl, True, 32)
    starts = Xs[ti[0]]
    lo, hi = np.full(d, -5.0), np.full(d, 5.0)
    Xp, fp, ne = eng.polish(starts, lo, hi, acq, pl, True, max_evals=50)
    assert Xp.shape == (32, d) and np.all(Xp >= lo) and np.all(Xp <= hi) and np.all(ne >= 1) and np.all(ne <= 50)
    assert np.all(fp >= tv[0] * (1 - 1e-12))  # never below its start
    # the value the polish reports IS the criterion at the point it reports
    _, _, _, _, vcheck, _ = eng.point_eval_batch(Xp, [acq], pl, True)
    np.testing.assert_allclose(vcheck[:, 0], fp, rtol=1e-12)

    def neg(x):
        mu, mse, dmu, dmse, v, dv = eng.point_eval_batch(x[None, :], [acq], pl, True)
        return -float(v[0, 0]), -dv[0, 0]

    fs = []
    for x0 in starts:
        x1, f1, _ = fmin_l_bfgs_b(neg, x0, pgtol=1e-8, factr=1e6, bounds=np.c_[lo, hi], maxfun=50)
        fs.append(-float(f1))
    fs = np.array(fs)
    assert fp.max() >= fs.max() * (1 - 1e-6), (fp.max(), fs.max())
    # start by start the lock-step optimiser is not systematically worse: at most a few starts end (slightly) lower
    worse = np.sum(fp < fs * (1 - 1e-3))
    assert worse <= 8, (worse, np.c_[fp, fs])


@pytest.mark.parametrize("hybrid,plain", [("sweep-BFGS", "sweep"), ("sweep-device-BFGS", "sweep-device")])
def test_sweep_bfgs_optimizer_uses_the_device_polish(eng, hybrid, plain):
    """`optimizer="sweep-BFGS"` / `"sweep-device-BFGS"` through `bogp.argmax_restart`: the sweep's top-k polished together;
    result >= the plain sweep's over the same candidates."""
    g = load_golden("G2_m32_ok_noisy")  # fmin's model: Matern-3/2, ordinary kriging
    d = g["X"].shape[1]
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    gp.set_state(g["par"], g["X"], g["y"])
    ei = bogp.EI(model=gp, minimize=True)
    box = bogp.optim.Box([(-5.0, 5.0)] * d, random_seed=3)
    np.random.seed(11)  # the device designs draw their Philox seed from the global stream
    x1, f1 = bogp.argmax_restart(ei, box, eval_budget=20000, optimizer=plain)
    box = bogp.optim.Box([(-5.0, 5.0)] * d, random_seed=3)
    np.random.seed(11)
    x2, f2 = bogp.argmax_restart(ei, box, eval_budget=20000, n_restart=16, optimizer=hybrid)
    assert len(x2) == d and f2 >= f1 and np.all(np.abs(x2) <= 5.0)
    np.testing.assert_allclose(np.ravel(ei(np.array([x2])))[0], f2, rtol=1e-9)


def test_large_batches_are_served_in_chunks(eng):
    """More rows than one launch takes (the right-hand sides of a chunk are capped at 256 MB / 4096 points): every row
    still equals its one-row answer, whatever chunk it fell into."""
    rng = np.random.default_rng(5)
    N, d, B = 300, 4, 5000
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(np.sin(X), axis=1).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    par = np.r_[np.full(d, 0.1), 0.9]
    eng.set_train(X, y)
    eng.commit(O.KERNEL_MATERN32, O.MODE_NOISY, par, 1e-5, True, 0.0)
    Xb = rng.uniform(-5, 5, size=(B, d))
    acq = [(O.ACQ_EI, 0.0), (O.ACQ_UCB, 0.5)]
    pl = float(y.min())
    mu, mse, dmu, dmse, vals, dvals = eng.point_eval_batch(Xb, acq, pl, True)
    assert mu.shape == (B,) and dvals.shape == (B, 2, d)
    for i in (0, 1, 4095, 4096, 4097, B - 1):
        m1, s1, a1, b1, v1 = eng.point_eval(Xb[i], acq, pl, True)
        np.testing.assert_allclose([m1, s1], [mu[i], mse[i]], rtol=1e-11, atol=1e-14)
        np.testing.assert_allclose(a1, dmu[i], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(v1, vals[i], rtol=1e-9, atol=1e-300)
    st = O.make_state(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-5, estimate_trend=True)
    omu, omse = O.predict(st, Xb)
    np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))


@pytest.mark.parametrize("N,d,kernel,est", [(200, 3, O.KERNEL_SE, True), (200, 3, O.KERNEL_SE, False), (700, 25, O.KERNEL_MATERN32, True), (1100, 11, O.KERNEL_MATERN52, True),
                                            (2048, 20, O.KERNEL_MATERN52, True), (513, 40, O.KERNEL_SE, False)])  # fmt: skip
def test_linear_trend_in_the_one_point_path(eng, N, d, kernel, est):
    """r05 (VERDICT r04 "missing" item 3): the linear basis f = [1, x] -- the one polynomial basis the reference's `gradient` can differentiate
    (gpr.py:556-575, trend.py:104-116) -- through the fused one-point / B-point kernels: mu, MSE, both input-gradients, criteria and their
    chain rule against the oracle, universal kriging (estimated coefficients) and fixed coefficients, one and several passes over the
    right-hand sides (d > 21), B = 1 (point in the kernel arguments) and a batch."""
    rng = np.random.default_rng(N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + 3.0 * X[:, 0] - X[:, d - 1] + rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.6 / d), 0.8]
    beta = None if est else np.linspace(-0.2, 0.3, d + 1)
    eng.set_train(X, y)
    eng.commit(kernel, O.MODE_NOISY, par, 1e-4, est, beta if beta is not None else 0.0, trend=O.TREND_LINEAR)
    st = O.make_state(par, X, y, kernel, O.MODE_NOISY, 1e-4, trend=O.TREND_LINEAR, estimate_trend=est, beta=beta)
    Xb = np.vstack([rng.uniform(-5, 5, size=(6, d)), X[:2] + 1e-2 * rng.standard_normal((2, d))])
    pl = O.plugin_value(st.y, True)
    acq = [(O.ACQ_EI, 0.0), (O.ACQ_UCB, 0.7), (O.ACQ_MGFI, 1.5)]
    mu, mse, dmu, dmse, vals, dvals = eng.point_eval_batch(Xb, acq, pl, True)
    omu, omse = O.predict(st, Xb)
    s2 = float(st.sigma2[0])
    np.testing.assert_allclose(mu, omu.ravel(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse.ravel(), rtol=1e-6, atol=1e-12 * s2)
    gp = bogp.GaussianProcess(corr="squared_exponential", thetaL=[1e-4] * d, thetaU=[1e2] * d)
    gp.sigma2, gp.y = st.sigma2, st.y
    for i in range(len(Xb)):
        odmu, odmse = O.gradient(st, Xb[i])
        np.testing.assert_allclose(dmu[i], np.ravel(odmu), rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(dmse[i], np.ravel(odmse), rtol=1e-6, atol=1e-9 * s2)
        for c, (aid, apar) in enumerate(acq):
            cls = {O.ACQ_EI: bogp.EI, O.ACQ_UCB: bogp.UCB, O.ACQ_MGFI: bogp.MGFI}[aid]
            kw = {O.ACQ_UCB: {"alpha": apar}, O.ACQ_MGFI: {"t": apar}}.get(aid, {})
            crit = cls(model=gp, minimize=True, **kw)
            mom = bogp.acquisition._Moments(crit, Xb[i : i + 1], (omu[i : i + 1], omse[i : i + 1], np.reshape(odmu, (-1, 1)), np.reshape(odmse, (-1, 1))))
            ov = O.acquisition(aid, apar, omu[i : i + 1], omse[i : i + 1], pl, s2, True)
            _, odx = crit._dx(mom, np.ravel(ov))
            np.testing.assert_allclose(vals[i, c], np.ravel(ov)[0], rtol=1e-6, atol=1e-300)
            np.testing.assert_allclose(dvals[i, c], np.ravel(odx), rtol=2e-6, atol=1e-9 * max(1e-300, float(np.max(np.abs(odx)))))
        m1, s1, a1, b1, v1 = eng.point_eval(Xb[i], acq, pl, True)  # B = 1
        np.testing.assert_allclose([m1, s1], [mu[i], mse[i]], rtol=1e-10, atol=1e-13)
        np.testing.assert_allclose(a1, dmu[i], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(b1, dmse[i], rtol=1e-9, atol=1e-12 * s2)
        # the plain (unfused) gradient entry point: the same numbers through other kernels
        g1, g2 = eng.gradient(Xb[i])
        np.testing.assert_allclose(g1, dmu[i], rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(g2, dmse[i], rtol=1e-7, atol=1e-10 * s2)
    # the sweep's posterior of the same rows (k_trend_terms / the producer-fused path): the trend mean is the same sum in the same order
    eng.upload_candidates(Xb)
    smu, smse = eng.predict()
    np.testing.assert_allclose(smu, mu, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(smse, mse, rtol=1e-8, atol=1e-12 * s2)
    with pytest.raises(_lib.BogpError):  # the quadratic basis has no Jacobian (trend.py:138-139): refused here as by bogp_gradient
        eng.commit(kernel, O.MODE_NOISY, par, 1e-4, True, 0.0, trend=O.TREND_QUADRATIC)
        eng.point_eval_batch(Xb, acq, pl, True)


@pytest.mark.parametrize("name", ["G13_linear_uk_se", "G15_linear_sk_se"])
def test_linear_trend_return_dx_matches_the_reference(eng, name):
    """The reference's own `gradient()` and `criterion(x, return_dx=True)` rows of a LINEAR-trend model (goldens G13: universal kriging, G15:
    fixed coefficients) through the fused B-point kernels, 1e-6."""
    g = load_golden(name)
    mode, kernel, est = int(g["mode"]), int(g["kernel"]), bool(g["estimate_trend"])
    nv = float(g["noise_var"][0]) if mode == O.MODE_NOISY else 0.0
    eng.set_train(g["X"], g["y"])
    eng.commit(kernel, mode, g["par"], nv, est, 0.0 if est else np.ravel(g["beta"]), trend=int(g["trend"]))
    nb = len(g["grad_mu"])
    Xb = g["Xs"][:nb]
    acq = [(i, p) for _, i, p in DX_ACQ]
    mu, mse, dmu, dmse, vals, dvals = eng.point_eval_batch(Xb, acq, float(g["plugin_eff"][0]), True)
    s2 = float(g["sigma2"][0])
    np.testing.assert_allclose(mu, g["mu"][:nb, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, g["mse"][:nb, 0], rtol=1e-6, atol=1e-12 * s2)
    np.testing.assert_allclose(dmu, g["grad_mu"][:, :, 0], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(dmse, g["grad_mse"][:, :, 0], rtol=1e-6, atol=1e-9 * s2)
    for c, (key, _, _) in enumerate(DX_ACQ):
        np.testing.assert_allclose(vals[:, c], g["dx_val_" + key], rtol=1e-6, atol=1e-300, err_msg=key)
        np.testing.assert_allclose(dvals[:, c, :], g["dx_" + key], rtol=1e-6, atol=1e-12, err_msg=key)


@pytest.mark.parametrize("N,d,tid", [(400, 100, O.TREND_CONSTANT), (300, 70, O.TREND_LINEAR), (500, 200, O.TREND_CONSTANT)])
def test_polish_beyond_64_dimensions_and_with_a_linear_trend(eng, N, d, tid):
    """r05: the lock-step polish for d up to BOGP_MAX_DIM (several coordinates per lane; r03-r04 refused d > 64) and on a
    linear-trend model.  Properties: stays in the box, never ends below its start, the reported value IS the criterion at the reported
    point, and it improves on the starts."""
    rng = np.random.default_rng(d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum((X - 1.0) ** 2, axis=1) + rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    par = np.r_[np.full(d, 0.3 / d), 0.8]
    eng.set_train(X, y)
    eng.commit(O.KERNEL_MATERN52, O.MODE_NOISY, par, 1e-4, True, 0.0, trend=tid)
    pl = float(y.min())
    acq = (O.ACQ_UCB, 0.5)
    starts = rng.uniform(-5, 5, size=(16, d))
    lo, hi = np.full(d, -5.0), np.full(d, 5.0)
    _, _, _, _, v0, _ = eng.point_eval_batch(starts, [acq], pl, True)
    Xp, fp, ne = eng.polish(starts, lo, hi, acq, pl, True, max_evals=40)
    assert Xp.shape == (16, d) and np.all(Xp >= lo) and np.all(Xp <= hi) and np.all(ne >= 1) and np.all(ne <= 40)
    assert np.all(fp >= v0[:, 0] - 1e-12 * np.abs(v0[:, 0]))
    assert np.mean(fp > v0[:, 0]) > 0.8  # it climbs
    _, _, _, _, vcheck, _ = eng.point_eval_batch(Xp, [acq], pl, True)
    np.testing.assert_allclose(vcheck[:, 0], fp, rtol=1e-12)
