"""GPU half of the round-2 host-integration rows: the fused q-criterion proposal behind ParallelBO's hook,
the nugget-retry path of `fit`, and the device-resident top-k."""
import numpy as np
import pytest

from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu

import bogp  # noqa: E402
from bogp import _lib, integration  # noqa: E402
from support.mini_driver import MiniParallelBO  # noqa: E402


def _fitted_model(N=300, d=4, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    gp = bogp.GaussianProcess(corr="matern52", thetaL=[1e-4] * d, thetaU=[1e2] * d, nugget=1e-6)
    gp.set_state(np.r_[np.full(d, 0.15), 0.9], X, y)
    return gp, X, y


class _Count:
    """Counts the posterior passes an engine makes (every one of these entry points walks all candidates once)."""

    def __init__(self, eng):
        self.n = 0
        for name in ("sweep", "sweep_topk", "predict"):
            orig = getattr(eng, name)

            def wrapped(*a, _o=orig, **kw):
                self.n += 1
                return _o(*a, **kw)

            setattr(eng, name, wrapped)


@pytest.mark.parametrize("fun,par", [("MGFI", {"t": 2}), ("UCB", {"alpha": 0.5})])
def test_fused_batch_is_one_pass_and_picks_what_q_separate_sweeps_would(fun, par):
    gp, X, y = _fitted_model()
    d, q, M = X.shape[1], 8, 20000
    hist = X[:50]
    drv = MiniParallelBO(gp, [(-5, 5)] * d, fun, par, "sweep", M, seed=9, history=hist)
    cnt = _Count(gp.engine)
    np.random.seed(4)
    xs, fs = integration.fused_batch_arg_max_acquisition(drv, q, False)
    assert cnt.n == 1  # ONE posterior pass for the q criteria
    assert len(xs) == q and len(fs) == q and all(len(x) == d for x in xs)
    # replay by hand: same t_i / alpha_i draws, same candidates, q full-value sweeps, then the selection rule
    np.random.seed(4)
    pars = [drv._sampler(drv._acquisition_par) for _ in range(q)]
    Xs = bogp.optim.Box([(-5, 5)] * d, random_seed=9).sample(M)
    eng = gp.engine
    eng.upload_candidates(Xs)
    acq_id = {"MGFI": _lib.ACQ_MGFI, "UCB": _lib.ACQ_UCB}[fun]
    pars_eff = [min(p, 22.36) if fun == "MGFI" else p for p in pars]
    _, _, vals = eng.sweep([(acq_id, p) for p in pars_eff], float(np.min(y)), True, return_values=True)
    taken = set()
    for c in range(q):
        order = np.lexsort((np.arange(M), -vals[c]))
        pick = next(i for i in order if i not in taken and not np.any(np.all(np.isclose(hist, Xs[i]), axis=1)))
        taken.add(int(pick))
        np.testing.assert_array_equal(np.asarray(xs[c]), Xs[pick])
        assert fs[c] == vals[c][pick]
    assert len(taken) == q
    # the oracle agrees on the values of the chosen rows
    st = O.make_state(np.r_[np.full(d, 0.15), 0.9], X, y, O.KERNEL_MATERN52, O.MODE_NOISY, 1e-6)
    mu, mse = O.predict(st, np.asarray(xs))
    for c in range(q):
        ref = O.acquisition(acq_id, pars_eff[c], mu[c : c + 1, 0], mse[c : c + 1, 0], float(np.min(y)), 0.9, True)[0]
        np.testing.assert_allclose(fs[c], ref, rtol=1e-6)


def test_fused_batch_on_device_generated_designs_and_other_optimisers_fall_through():
    gp, X, y = _fitted_model(N=200, d=3)
    drv = MiniParallelBO(gp, [(-5, 5)] * 3, "MGFI", {"t": 2}, "sweep-device-lhs", 30000, seed=1)
    cnt = _Count(gp.engine)
    np.random.seed(0)
    xs, fs = integration.fused_batch_arg_max_acquisition(drv, 4, False)
    assert cnt.n == 1 and len({tuple(x) for x in xs}) == 4
    assert all(np.all(np.abs(np.asarray(x)) <= 5) for x in xs) and all(np.isfinite(fs))
    # an optimiser the hook does not serve is handed to the original method untouched
    integration._ORIGINAL["batch"] = lambda self, n, r, f=None: ("orig", n)
    try:
        drv._optimizer = "BFGS"
        assert integration.fused_batch_arg_max_acquisition(drv, 4, True) == ("orig", 4)
    finally:
        integration._ORIGINAL.clear()


def test_topk_is_repeated_argmax_for_every_criterion():
    gp, X, y = _fitted_model(N=150, d=3)
    eng = gp.engine
    rng = np.random.default_rng(0)
    Xs = rng.uniform(-5, 5, size=(5000, 3))
    Xs[100] = Xs[7]  # an exact tie between two candidates: the lower index ranks first
    eng.upload_candidates(Xs)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0), (_lib.ACQ_UCB, 0.5)]
    pl = float(y.min())
    _, _, vals = eng.sweep(acq, pl, True, return_values=True)
    tv, ti = eng.sweep_topk(acq, pl, True, 32)
    for c in range(3):
        order = np.lexsort((np.arange(len(Xs)), -vals[c]))[:32]
        np.testing.assert_array_equal(ti[c], order)
        np.testing.assert_array_equal(tv[c], vals[c][order])
    # fewer candidates than k: padded with (-inf, -1)
    eng.upload_candidates(Xs[:5])
    tv, ti = eng.sweep_topk(acq[:1], pl, True, 8)
    assert np.all(ti[0, 5:] == -1) and np.all(np.isneginf(tv[0, 5:])) and sorted(ti[0, :5]) == [0, 1, 2, 3, 4]


def _smooth(n=40, d=2, seed=2):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(n, d))
    y = np.sum(X**2, axis=1)
    return X, ((y - y.mean()) / y.std()).reshape(-1, 1)


@pytest.mark.parametrize("likelihood", ["concentrated", "restricted"])
def test_nugget_retry_switches_noiseless_to_noisy_on_the_device(likelihood):
    """gpr.py:384-399 on smooth data whose noiseless likelihood is rejected everywhere in the box (llf > 0 or a failed
    Cholesky): the fit ends in the noisy mode with a grown nugget and a state of THAT mode's parameter layout."""
    X, y = _smooth()
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(2) if likelihood == "restricted" else None, corr="squared_exponential",
                              thetaL=[1e-2] * 2, thetaU=[1.0] * 2, nugget=0, random_start=2, eval_budget=60, likelihood=likelihood)  # fmt: skip
    assert gp.estimation_mode == "noiseless"
    np.random.seed(0)
    gp.fit(X, y)
    assert gp.is_fitted and len(gp._committed_par) == 3 and np.isfinite(gp.log_likelihood_)
    if likelihood == "concentrated":  # (the REML value of the same data is finite without a nugget: that fit stays noiseless,
        assert gp.estimation_mode == "noisy" and float(np.ravel(gp.noise_var)[0]) >= 1e-5  # with [theta, sigma2] as well)
    mu, mse = gp.predict(X[:4], eval_MSE=True)
    assert np.all(np.isfinite(mu)) and np.all(mse >= 0)
    if likelihood == "concentrated":
        st = O.make_state(gp._committed_par, X, y, O.KERNEL_SE, O.MODE_NOISY, float(np.ravel(gp.noise_var)[0]))
        np.testing.assert_allclose(gp.log_likelihood_, st.llf, rtol=1e-9)
        omu, omse = O.predict(st, X[:4])
        np.testing.assert_allclose(mu, omu, rtol=1e-6, atol=1e-9)


def test_likelihood_calls_leave_the_fitted_device_state_alone():
    gp, X, y = _fitted_model(N=120, d=3)
    Xs = np.random.default_rng(1).uniform(-5, 5, size=(64, 3))
    par0 = gp._committed_par.copy()
    mu0, mse0 = gp.predict(Xs, eval_MSE=True)
    other = par0 * np.array([3.0, 0.3, 2.0, 1.0])
    env = {}
    assert np.isfinite(gp.log_likelihood_concentrated(other, env)) and "C" in env
    gp.log_likelihood_concentrated(other, eval_grad=True)
    gp.log_likelihood_concentrated(other, {}, eval_grad=True)
    np.testing.assert_array_equal(gp._committed_par, par0)
    mu1, mse1 = gp.predict(Xs, eval_MSE=True)
    np.testing.assert_array_equal(mu1, mu0)
    np.testing.assert_array_equal(mse1, mse0)


def test_set_train_reuses_its_buffers_across_a_growing_training_set():
    """A BO loop adds points one tell() at a time: every size must give the state a fresh engine gives (the N x N
    buffers are kept between bogp_set_train calls that fit their capacity)."""
    rng = np.random.default_rng(5)
    d = 3
    Xall = rng.uniform(-5, 5, size=(400, d))
    yall = np.sum(Xall**2, axis=1, keepdims=True)
    yall = (yall - yall.mean()) / yall.std() + 0.3 * rng.standard_normal((400, 1))
    Xs = rng.uniform(-5, 5, size=(100, d))
    par = np.r_[np.full(d, 0.15), 0.9]
    e1 = _lib.Engine(0)
    for N in (300, 301, 320, 333, 257, 64, 65, 400, 129):
        e1.set_train(Xall[:N], yall[:N])
        l1 = e1.commit(3, 1, par, 1e-6)
        e1.upload_candidates(Xs)
        m1, s1 = e1.predict()
        g1 = e1.nll(3, 1, par, 1e-6, eval_grad=True)
        e2 = _lib.Engine(0)
        e2.set_train(Xall[:N], yall[:N])
        l2 = e2.commit(3, 1, par, 1e-6)
        e2.upload_candidates(Xs)
        m2, s2 = e2.predict()
        g2 = e2.nll(3, 1, par, 1e-6, eval_grad=True)
        e2.close()
        assert l1 == l2 and g1[0] == g2[0]
        np.testing.assert_array_equal(m1, m2)
        np.testing.assert_array_equal(s1, s2)
        np.testing.assert_array_equal(g1[1], g2[1])
    e1.close()


@pytest.mark.parametrize("N,d,kernel,mode,est", [
    (33, 1, O.KERNEL_SE, O.MODE_NOISY, False), (100, 5, O.KERNEL_MATERN32, O.MODE_NOISY, True), (128, 4, O.KERNEL_MATERN52, O.MODE_NOISY, False),
    (129, 7, O.KERNEL_SE, O.MODE_NOISY, True), (200, 6, O.KERNEL_MATERN52, O.MODE_NOISY, True),
    (256, 10, O.KERNEL_SE, O.MODE_NOISY, False), (257, 3, O.KERNEL_MATERN32, O.MODE_NOISY, True),
    (500, 20, O.KERNEL_ABSEXP, O.MODE_NOISY, False), (512, 10, O.KERNEL_SE, O.MODE_NOISY, True),
    (480, 64, O.KERNEL_MATERN12, O.MODE_NOISY, False),
])  # fmt: skip
def test_fused_small_sweep_equals_the_chunked_schedule_and_the_oracle(N, d, kernel, mode, est):
    """k_sweep_small (N <= 512: producer + contraction + criteria + argmax in one launch, r resident in LDS) against the
    chunked three-kernel schedule of the same library (BOGP_NO_FUSED_SMALL=1) and against the oracle, ragged M included."""
    import os

    rng = np.random.default_rng(N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    theta = np.full(d, 0.4 / d) * rng.uniform(0.7, 1.3, size=d)
    par = np.r_[theta, 0.9 if mode == O.MODE_NOISY else 0.98]
    nv = 1e-6 if mode == O.MODE_NOISY else 0.0
    st = O.make_state(par, X, y, kernel, mode, nv, estimate_trend=est, beta=0.0)
    eng = _lib.Engine(0)
    eng.set_train(X, y)
    eng.commit(kernel, mode, par, nv, est, 0.0)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0), (_lib.ACQ_UCB, 0.5), (_lib.ACQ_EPSILON_PI, 1e-10)]
    pl = float(y.min())
    for M in (33, 63, 64, 65, 1000, 4097, 16384 + 700, 16384 + 5000):  # the last two: bulk launch + 32- / 48-candidate tail launch
        Xs = rng.uniform(-5, 5, size=(M, d))
        Xs[M // 2] = X[3]  # a candidate on a training point: MSE at nugget level, guards in play
        eng.upload_candidates(Xs)
        out = {}
        for tag, flag in (("fused", "0"), ("chunked", "1")):
            os.environ["BOGP_NO_FUSED_SMALL"] = flag
            try:
                mu, mse = eng.predict()
                mu_only, _ = eng.predict(eval_MSE=False)
                best, idx, vals = eng.sweep(acq, pl, True, return_values=True)
                b2, i2 = eng.sweep(acq, pl, True)
            finally:
                del os.environ["BOGP_NO_FUSED_SMALL"]
            assert eng.last_timing()["n_chunks"] == 1
            np.testing.assert_array_equal(mu_only, mu)
            np.testing.assert_array_equal(i2, idx)
            np.testing.assert_array_equal(b2, best)
            for c in range(len(acq)):
                assert idx[c] == int(np.argmax(vals[c])) and best[c] == vals[c][idx[c]]
            out[tag] = (mu, mse, vals, idx)
        omu, omse = O.predict_chunked(st, Xs, 512)
        for tag in out:
            np.testing.assert_allclose(out[tag][0], omu[:, 0], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(out[tag][1], omse[:, 0], rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))
        # two summation orders of the same sums (mu = sum r gamma cancels heavily when gamma is large: compare on its scale)
        np.testing.assert_allclose(out["fused"][0], out["chunked"][0], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(out["fused"][1], out["chunked"][1], rtol=1e-7, atol=1e-12 * float(st.sigma2[0]))
        solid = omse[:, 0] > 1e-9 * float(st.sigma2[0])  # away from training points the criteria are well conditioned
        for c, (a_id, a_par) in enumerate(acq):
            ref = O.acquisition(a_id, a_par, omu[:, 0], omse[:, 0], pl, float(st.sigma2[0]), True)
            np.testing.assert_allclose(out["fused"][2][c][solid], ref[solid], rtol=1e-6, atol=1e-300)
            if np.all(solid) or int(np.argmax(ref)) != M // 2:
                assert out["fused"][3][c] == int(np.argmax(ref)) == out["chunked"][3][c]
    eng.close()


def test_library_exchange_with_a_one_rank_communicator():
    """bogp_comm_init + bogp_exchange_argmax / bogp_exchange_topk on a one-rank RCCL communicator (all a 1-GPU box can
    run): the queued sweep (no host read-back) followed by the exchange returns the sweep's own winners with the index
    offset applied and the winning points attached; the top-k flavour likewise."""
    from bogp import distributed

    gp, X, y = _fitted_model(N=700, d=5)  # N > 512: the chunked schedule; the fused one is covered below
    eng = gp.engine
    assert distributed.init_engine_comm(eng) == (0, 1) and eng.comm_info() == (0, 1)
    rng = np.random.default_rng(3)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0), (_lib.ACQ_UCB, 0.5)]
    pl = float(y.min())
    for n_train in (700, 300):
        if n_train != 700:
            gp2, X2, y2 = _fitted_model(N=n_train, d=5)
            eng = gp2.engine
            distributed.init_engine_comm(eng)
            pl = float(y2.min())
        Xs = rng.uniform(-5, 5, size=(3000, 5))
        eng.upload_candidates(Xs)
        best, idx = eng.sweep(acq, pl, True)
        assert eng.sweep(acq, pl, True, local_result=False) is None
        gv, gi, gx = eng.exchange_argmax(len(acq), 10_000_000_000, True)
        np.testing.assert_array_equal(gv, best)
        np.testing.assert_array_equal(gi, idx + 10_000_000_000)
        np.testing.assert_array_equal(gx, Xs[idx])
        tv, ti = eng.sweep_topk(acq, pl, True, 7)
        ev, ei, ex = eng.exchange_topk(len(acq), 7, 123, True)
        np.testing.assert_array_equal(ev, tv)
        np.testing.assert_array_equal(ei, ti + 123)
        np.testing.assert_array_equal(ex, Xs[ti])
        # stale winners are refused: a top-k sweep overwrites the argmax buffers; new candidates change the rows
        with pytest.raises(ValueError):
            eng.exchange_argmax(len(acq), 0, True)
        with pytest.raises(ValueError):
            eng.exchange_topk(len(acq), 5, 0, True)  # not the shape the last sweep_topk produced
        eng.upload_candidates(Xs[:100])
        with pytest.raises(ValueError):
            eng.exchange_topk(len(acq), 7, 0, True)
        bv, bi = np.empty(256), np.empty(256, dtype=np.int64)
        for fn in (eng._lib.bogp_exchange_argmax, eng._lib.bogp_exchange_topk):  # and by the C side itself
            assert fn(eng._h, 0, _lib._ptr(bv), bi.ctypes.data_as(_lib._lp), None) == _lib.ERR_INVALID
            assert b"no " in eng._lib.bogp_last_error(eng._h)
        eng.upload_candidates(Xs)
        # the optimiser front ends pick the library transport up by themselves
        crit = [bogp.MGFI(model=gp if n_train == 700 else gp2, t=1.0), bogp.MGFI(model=gp if n_train == 700 else gp2, t=3.0)]
        v, g, x = bogp.sweep_argmax(crit, Xs, index_offset=50)
        assert np.all(g >= 50) and np.array_equal(x, Xs[g - 50])
    eng.comm_destroy()
    assert eng.comm_info() == (0, 0)


@pytest.mark.parametrize("N,d", [(17, 2), (64, 3), (128, 5), (130, 5), (200, 32), (256, 10), (250, 33)])
def test_fused_small_sweep_tiles_per_wave_variants_are_bit_identical(N, d):
    """r03: up to N = 256 k_sweep_small runs with FOUR waves per workgroup and two workgroups per CU (each wave carries the sums
    of two waves of the eight-wave schedule, separately and in the same order), or -- BOGP_SMALL_NW=8 -- with eight waves that own
    two column tiles of V instead of four and an 8-slot B-fragment ring.  Same tiles, same k order, same summation order: the
    outputs must be the SAME BITS as with the r02 schedule (BOGP_SMALL_NW=8 BOGP_SMALL_NR=4), values and argmax alike."""
    import os

    rng = np.random.default_rng(1000 * N + d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(np.cos(X), axis=1)
    y = ((y - y.mean()) / y.std() + 0.05 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.3 / d), 0.9]
    eng = _lib.Engine(0)
    eng.set_train(X, y)
    eng.commit(O.KERNEL_MATERN52, O.MODE_NOISY, par, 1e-6, True, 0.0)
    acq = [(_lib.ACQ_EI, 0.0), (_lib.ACQ_MGFI, 2.0)]
    for M in (64, 1000, 16384, 32768, 16384 + 700):  # (the four-wave schedule is taken from 257 workgroups on)
        eng.upload_candidates(rng.uniform(-5, 5, size=(M, d)))
        out = []
        for env in ({}, {"BOGP_SMALL_NW": "8"}, {"BOGP_SMALL_NW": "8", "BOGP_SMALL_NR": "4"}):
            os.environ.update(env)
            try:
                mu, mse = eng.predict()
                mu_only, _ = eng.predict(eval_MSE=False)
                best, idx, vals = eng.sweep(acq, float(y.min()), True, return_values=True)
            finally:
                for k in env:
                    os.environ.pop(k, None)
            out.append((mu, mse, mu_only, best, idx, vals))
        for other in out[1:]:
            for a, b in zip(out[0], other):
                if M % 16384 == 0 or M < 16384:
                    np.testing.assert_array_equal(a, b)
                else:  # the eight-wave schedule serves the last, incomplete round with 32- / 48-candidate workgroups, whose
                    # instantiations round a few sums differently from the 64-candidate one (1e-15): same numbers, not same bits
                    np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13)
    eng.close()


def test_mle_restarts_on_several_streams_of_one_gpu():
    """GaussianProcess(restart_streams=T): the restarts of the MLE (gpr.py:1127-1162) run on T engines (= HIP streams) of the same GPU
    at once, every worker with 1 / T of the budget, all starting points from the global np.random stream in the sequential order
    (the one-device flavour of distribute_restarts).  Opt-in because it relaxes the shared budget / stagnation counter.
    Checked: same starting points as the sequential run (the first restart's result is reproduced exactly by worker 0), a likelihood at
    least as good as the sequential run's first restart, a valid committed state (posterior against the oracle at the fitted
    parameters), deterministic results, and the evaluation count within the budget."""
    rng = np.random.default_rng(5)
    N, d = 90, 4
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(np.sin(X), axis=1) + 0.1 * rng.standard_normal(N)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    width = np.full(d, 10.0)

    def make(streams, random_start):
        return bogp.GaussianProcess(mean=bogp.trend.constant_trend(d), corr="matern", thetaL=1e-3 * width, thetaU=1e2 * width, nugget=1e-6,
                                    optimizer="BFGS", wait_iter=3, random_start=random_start, eval_budget=400, restart_streams=streams)

    np.random.seed(11)
    one = make(1, 1).fit(X, y)  # one restart from the first starting point
    llf_first = float(one.log_likelihood_concentrated(np.r_[one.theta_, one.sigma2]))
    fits = []
    for _ in range(2):
        np.random.seed(11)
        gp = make(4, 8).fit(X, y)
        fits.append(gp)
    a, b = fits
    np.testing.assert_array_equal(a.theta_, b.theta_)
    assert a.eval_count == b.eval_count and 0 < a.eval_count <= 400 + 4 * 8  # (L-BFGS-B may overshoot maxfun by a line search)
    llf_multi = float(a.log_likelihood_concentrated(np.r_[a.theta_, a.sigma2]))
    assert llf_multi >= llf_first - 1e-9 * abs(llf_first)  # worker 0 ran exactly that restart with a quarter of the budget... or found better
    Xs = rng.uniform(-5, 5, size=(300, d))
    mu, mse = a.predict(Xs, eval_MSE=True)
    st = O.make_state(np.r_[a.theta_, a.sigma2], X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, estimate_trend=True)
    omu, omse = O.predict(st, Xs)
    np.testing.assert_allclose(mu, omu, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse, omse, rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))


@pytest.mark.parametrize("N", [3300, 4096, 6144, 6200, 7000])
def test_large_fit_path_equals_the_64_block_path(N):
    """Above the elimination's limit (N > 3072; from ld = 6144 on until the LDS-free tile core of r06) the inverse and R^-1 = U U^T run on 128 x 128
    tiles (k_mm128, kernels_chol.hip), and from ld = 6144 on the Cholesky's first block columns as WIDE PANELS (r06: rank-64 updates confined to the panel, one rank-64 w k_mm128 update of the rest; by default 24 block columns a
    panel while 72 stay behind; BOGP_BIG_CHOL=0: none, a list: that schedule).  Same mathematics, other summation order than the 64-block path: likelihood, gradient,
    committed state and posterior against the 64-block path (BOGP_NO_BIG_FIT=1), incl. sizes whose tile count is not a power of two, schedules
    with panels of unequal width, and the -inf convention on an indefinite matrix.  Among themselves the schedules are BIT-identical."""
    import ctypes
    import os

    d = 6
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std() + 0.3 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.2) * rng.uniform(0.8, 1.2, size=d), 0.9]
    Xs = rng.uniform(-5, 5, size=(300, d))
    out = {}
    variants = {"default": (None, None), "small": ("1", None), "one-level": (None, "0"), "one panel": (None, "32"), "three panels": (None, "16,10,6")}
    for tag, (no_big, sched) in variants.items():
        for k, v in (("BOGP_NO_BIG_FIT", no_big), ("BOGP_BIG_CHOL", sched)):
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        try:
            eng = _lib.Engine(0)
            eng.set_train(X, y)
            llf, grad = eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True)
            llf_c = eng.commit(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0)
            st = eng.get_state()
            eng.upload_candidates(Xs)
            mu, mse = eng.predict()
            out[tag] = (llf, grad, llf_c, st, mu, mse)
            if tag != "small":  # an indefinite matrix (negative nugget: diagonal below the off-diagonal mass) is reported as LAPACK would
                with pytest.raises(_lib.NotPositiveDefinite):
                    eng.commit(_lib.KERNEL_SE, _lib.MODE_NOISY, np.r_[np.full(d, 1e-4), 0.9], -0.6)
            eng.close()
        finally:
            os.environ.pop("BOGP_NO_BIG_FIT", None)
            os.environ.pop("BOGP_BIG_CHOL", None)
    for variant in variants:
        if variant != "small":
            _compare_fit_outputs(out[variant], out["small"])
    # every schedule is the one-level chain's arithmetic in another order of LAUNCHES, not of operations: a wide panel's rank-64 w product starts from
    # the tile it updates and runs over k in the order of the w rank-64 updates it replaces -- the factor, and everything computed from it, is the same bits
    w = (ctypes.c_int * 16)()
    if N >= 6017:  # (the default does take a wide panel at these sizes; below, the explicit schedules of this test still do)
        assert _lib.load().bogp_chol_wide_panels(N, w, 16) == 1 and w[0] == 24
    else:
        assert _lib.load().bogp_chol_wide_panels(N, w, 16) == 0
    for variant in ("default", "one panel", "three panels"):
        np.testing.assert_array_equal(out[variant][3]["C"], out["one-level"][3]["C"])
        assert out[variant][0] == out["one-level"][0] and out[variant][2] == out["one-level"][2]
        np.testing.assert_array_equal(out[variant][1], out["one-level"][1])
        np.testing.assert_array_equal(out[variant][5], out["one-level"][5])


@pytest.mark.parametrize("N,d,trend", [(3300, 3, 1), (3585, 2, 2)])
def test_large_fit_path_with_a_polynomial_trend_and_the_restricted_likelihood(N, d, trend):
    """The 128-tile path (every N > 3072 since r06) under a linear / quadratic basis (trend.py:94-142) and for the restricted likelihood
    (gpr.py:813-918): against the 64-block kernels (BOGP_NO_BIG_FIT=1; the factor, and with it llf / REML / mu, are the same bits) and
    against the oracle, incl. the posterior of a trend sweep (k_mm128's chunk products)."""
    import os

    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1) + X[:, 0]
    y = ((y - y.mean()) / y.std() + 0.2 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.3), 0.8]
    Xs = rng.uniform(-5, 5, size=(777, d))
    out = {}
    for tag, flag in (("big", None), ("small", "1")):
        os.environ.pop("BOGP_NO_BIG_FIT", None)
        if flag:
            os.environ["BOGP_NO_BIG_FIT"] = flag
        try:
            eng = _lib.Engine(0)
            eng.set_train(X, y)
            llf, grad = eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=trend)
            rl, rg = eng.nll_restricted(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=trend)
            eng.commit(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, trend=trend)
            eng.upload_candidates(Xs)
            mu, mse = eng.predict()
            out[tag] = (llf, grad, rl, rg, mu, mse)
            eng.close()
        finally:
            os.environ.pop("BOGP_NO_BIG_FIT", None)
    b, s = out["big"], out["small"]
    assert b[0] == s[0] and b[2] == s[2]
    np.testing.assert_array_equal(b[4], s[4])
    np.testing.assert_allclose(b[1], s[1], rtol=1e-8, atol=1e-9 * np.abs(s[1]).max())
    np.testing.assert_allclose(b[3], s[3], rtol=1e-8, atol=1e-9 * np.abs(s[3]).max())
    ollf = float(O.log_likelihood_concentrated(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True))
    orl = float(O.log_likelihood_restricted(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True))
    np.testing.assert_allclose([b[0], b[2]], [ollf, orl], rtol=1e-9)
    st = O.make_state(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True)
    omu, omse = (np.asarray(v).ravel() for v in O.predict(st, Xs))
    np.testing.assert_allclose(b[4], omu, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(b[5], omse, rtol=1e-6, atol=1e-12 * float(st.sigma2[0]))


def _compare_fit_outputs(b, s):
    np.testing.assert_allclose(b[0], s[0], rtol=1e-11)
    np.testing.assert_allclose(b[2], s[2], rtol=1e-11)
    np.testing.assert_allclose(b[1], s[1], rtol=1e-8, atol=1e-9 * np.abs(s[1]).max())
    scale = np.abs(s[3]["C"]).max()
    np.testing.assert_allclose(b[3]["C"], s[3]["C"], rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(b[3]["gamma"], s[3]["gamma"], rtol=1e-6, atol=1e-8 * np.abs(s[3]["gamma"]).max())
    np.testing.assert_allclose(b[4], s[4], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(b[5], s[5], rtol=1e-6, atol=1e-11)


@pytest.mark.parametrize("name", ["G25_cubic_ok_noisy", "G26_genexp_sk_noisy", "G31_matern_nu08_ok_noisy", "G32_matern_nu37_sk_noisy"])
def test_value_only_kernels_on_the_device(name):
    """cubic / generalized_exponential / the general-nu Matern arm (kernel.py:201-207; a K_nu of real order on the device where the reference
    calls scipy.special.kv) against the reference's outputs (G25 / G26 / G31 / G32): committed state, posterior, the six
    criteria with argmax, likelihood values incl. the -inf convention; derivatives are refused (the reference has none)."""
    from conftest import load_golden, state_from_golden

    g = load_golden(name)
    st = state_from_golden(g)
    kid, mode = int(g["kernel"]), int(g["mode"])
    est = bool(g["estimate_trend"])
    eng = _lib.Engine(0)
    eng.set_train(g["X"], g["y"])
    llf = eng.commit(kid, mode, g["par"], 1e-6, est, 0.0)
    np.testing.assert_allclose(llf, float(g["llf"]), rtol=1e-10)
    s = eng.get_state()
    np.testing.assert_allclose(s["C"], g["C"], rtol=0, atol=1e-11 * np.abs(g["C"]).max())
    np.testing.assert_allclose(s["gamma"], g["gamma"].ravel(), rtol=1e-6, atol=1e-9 * np.abs(g["gamma"]).max())
    acq = [("EI", O.ACQ_EI, 0.0), ("EpsilonPI_1e-10", O.ACQ_EPSILON_PI, 1e-10), ("UCB_0.5", O.ACQ_UCB, 0.5),
           ("MGFI_1", O.ACQ_MGFI, 1.0), ("MGFI_2", O.ACQ_MGFI, 2.0), ("MGFI_100", O.ACQ_MGFI, 100.0)]  # fmt: skip
    pl = O.plugin_value(st.y, True)
    import os

    for fused in ("0", "1"):  # both schedules of the sweep (N = 70: the fused small-N kernel by default)
        os.environ["BOGP_NO_FUSED_SMALL"] = fused
        try:
            eng.upload_candidates(g["Xs"])
            mu, mse = eng.predict()
            best, idx, vals = eng.sweep([(a, p) for _, a, p in acq], pl, True, return_values=True)
            eng.upload_candidates(g["Xs"][:20])  # the small-batch path (M <= 32)
            mu20, mse20 = eng.predict()
        finally:
            del os.environ["BOGP_NO_FUSED_SMALL"]
        np.testing.assert_allclose(mu, g["mu"][:, 0], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mse, g["mse"][:, 0], rtol=1e-6, atol=1e-12 * st.sigma2[0])
        np.testing.assert_allclose(mu20, g["mu"][:20, 0], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mse20, g["mse"][:20, 0], rtol=1e-6, atol=1e-12 * st.sigma2[0])
        solid = g["mse"][:, 0] > 1e-9 * st.sigma2[0]
        for c, (key, a, p) in enumerate(acq):
            np.testing.assert_allclose(vals[c][solid], g[key][solid], rtol=1e-6, atol=1e-300)
            if int(g["argmax_" + key][0]) != 5:
                assert idx[c] == int(g["argmax_" + key][0])
    for mid in (0, 1, 2):
        for tname, e_ in (("sk", False), ("ok", True)):
            P, L = g["t_m%d_%s_par" % (mid, tname)], g["t_m%d_%s_llf" % (mid, tname)]
            for p_, l_ in zip(P, L):
                if np.isfinite(l_):
                    try:
                        np.testing.assert_allclose(eng.nll(kid, mid, p_, 1e-6 if mid == 1 else 0.0, e_, 0.0), l_, rtol=1e-8)
                    except _lib.NotPositiveDefinite:
                        # only acceptable where the matrix is singular to working precision (the noiseless cubic tables hold
                        # such cases: LAPACK got through on a pivot of rounding size, a different summation order does not)
                        n_th = len(p_) - (0 if mid == 0 else 1)
                        R0 = O.correlation_matrix(kid, p_[:n_th], g["X"])
                        assert mid == 0 and np.linalg.cond(R0) > 1e13, (mid, tname, np.linalg.cond(R0))
                else:
                    with pytest.raises(_lib.NotPositiveDefinite):
                        eng.nll(kid, mid, p_, 1e-6 if mid == 1 else 0.0, e_, 0.0)
    eng.commit(kid, mode, g["par"], 1e-6, est, 0.0)
    for call in (lambda: eng.nll(kid, mode, g["par"], 1e-6, est, 0.0, eval_grad=True), lambda: eng.gradient(g["Xs"][0]),
                 lambda: eng.point_eval(g["Xs"][0]), lambda: eng.gradient_batch(g["Xs"][:3])):  # fmt: skip
        with pytest.raises(_lib.BogpError) as ei:
            call()
        assert ei.value.code == _lib.ERR_UNSUPPORTED
    # the drop-in class: pinned state, predictions and a sweep through the front end
    d = g["X"].shape[1]
    user_par = g["par"]
    if kid == 7:  # the order is a keyword of the correlation function for the user, a trailing theta entry for the engine
        import functools

        def matern(theta, X, nu=1.5):  # (only its name and keyword are looked at: surrogate.kernel_id_of)
            raise AssertionError("the device path never calls the correlation function")

        corr = functools.partial(matern, nu=float(g["nu"]))
        user_par = np.r_[g["par"][:d], g["par"][d + 1:]]
        n = d
    else:
        corr = {5: "cubic", 6: "generalized_exponential"}[kid]
        n = len(g["par"]) - 1
    gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d, beta=None if est else 0.0), corr=corr, thetaL=[1e-5] * n, thetaU=[1e2] * n,
                              nugget=1e-6)  # fmt: skip
    gp.set_state(user_par, g["X"], g["y"])
    if kid == 7:
        with pytest.raises(NotImplementedError):
            gp.fit(g["X"], g["y"])
        gp.set_state(user_par, g["X"], g["y"])
    m2, s2 = gp.predict(g["Xs"], eval_MSE=True)
    np.testing.assert_allclose(m2, g["mu"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(np.ravel(bogp.UCB(model=gp)(g["Xs"])), g["UCB_0.5"], rtol=1e-6)
    eng.close()


def test_a_failed_factorisation_does_not_poison_the_next_one():
    """Regression (found with G25): a factorisation that breaks down (singular to working precision: pivots of rounding
    size, overflowing block inverses, inf * 0) used to leave NaN in the identity padding of the in-place factor (N not a
    multiple of 64), and every later likelihood on the handle then failed too -- inside an MLE that turned one bad
    L-BFGS-B step into a dead restart."""
    rng = np.random.default_rng(8)
    N, d = 70, 3
    X = rng.uniform(-5, 5, size=(N, d))
    X[N - 1] = X[0] + 1e-13  # numerically duplicated rows: singular without a nugget
    y = np.sum(X**2, axis=1, keepdims=True)
    y = (y - y.mean()) / y.std() + 0.1 * rng.standard_normal((N, 1))
    good = np.r_[np.full(d, 0.2), 0.9]
    eng = _lib.Engine(0)
    eng.set_train(X, y)
    ref = eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISY, good, 1e-3, False, 0.0, eval_grad=True)
    for bad_call in (lambda: eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISELESS, np.full(d, 1e-9), 0.0, False, 0.0),
                     lambda: eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISY, good, -0.95, False, 0.0),
                     lambda: eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISY, np.r_[np.full(d, 1e-12), 0.9], 0.0, False, 0.0)):  # fmt: skip
        try:
            bad_call()
        except _lib.NotPositiveDefinite:
            pass
        again = eng.nll(_lib.KERNEL_SE, _lib.MODE_NOISY, good, 1e-3, False, 0.0, eval_grad=True)
        assert again[0] == ref[0]
        np.testing.assert_array_equal(again[1], ref[1])
    eng.close()


def test_device_designs_apply_scale_and_precision_like_realspace_sample():
    """search_space.py:754 `self.round(self.to_linear_scale(X))` on the device: the three designs drawn in the transformed
    box, mapped back per variable (log10 / logit / bilog / log / linear) and rounded + clipped where a precision is set --
    against the NumPy restatement (oracle/philox.py): exact where only rounding is involved, within 2 ulp through exp / pow
    (the device's libm against NumPy's)."""
    from bogp import optim
    from oracle import philox as P

    bounds = [(1e-3, 10.0), (-5.0, 5.0), (0.01, 0.99), (-100.0, 100.0), (0.5, 50.0), (-1.0, 1.0)]
    scale = ["log10", "linear", "logit", "bilog", "log", None]
    prec = [None, 2, None, 1, 3, None]
    box = optim.Box(bounds, random_seed=0, precision=prec, scale=scale)
    lo_t, hi_t, scales, precs, lo, hi = optim.design_of(box)
    d = len(bounds)
    eng = _lib.Engine(0)
    eng.set_train(np.zeros((4, d)) + np.arange(4)[:, None], np.arange(4.0))
    M = 20000
    for method, ref in (("uniform", lambda: P.uniform_box(lo_t, hi_t, M, 77, first_row=123)),
                        ("LHS", lambda: P.lhs_box(lo_t, hi_t, M, 77, first_row=123, n_strata=M + 123)),
                        ("sobol", lambda: P.sobol_box(lo_t, hi_t, M, _lib.sobol_direction_numbers(d), first_index=124))):  # fmt: skip
        eng.set_candidate_transform(scales, precs, lo, hi)
        eng.generate_candidates(lo_t, hi_t, M, seed=77, first_row=123, method=method, n_total=M + 123)
        got = eng.read_candidates(np.arange(M))
        want = P.transform(ref(), scales, precs, lo, hi)
        for k in range(d):
            assert np.all(got[:, k] >= lo[k]) and np.all(got[:, k] <= hi[k])
            if scales[k] == "linear":
                np.testing.assert_array_equal(got[:, k], want[:, k])  # Philox integers + a separately rounded multiply-add + np.round
            elif precs[k] is None:
                np.testing.assert_allclose(got[:, k], want[:, k], rtol=5e-16, atol=0)
            else:  # rounded after a transcendental: equal except where 2 ulp straddle a rounding boundary
                differ = got[:, k] != want[:, k]
                assert differ.mean() < 1e-3 and np.all(np.abs(got[differ, k] - want[differ, k]) <= 10.0 ** -precs[k] * 1.0000001)
        # the setting is sticky until reset
        eng.set_candidate_transform()
        eng.generate_candidates(lo_t, hi_t, 16, seed=77, method=method)
        plain = eng.read_candidates(np.arange(16))
        assert np.all(plain >= lo_t - 1e-12) and np.all(plain <= hi_t + 1e-12)
    eng.close()
    # through the optimiser front end: the proposal honours precision and bounds of the space
    gp, X, y = _fitted_model(N=120, d=3)
    sp = optim.Box([(-5, 5)] * 3, random_seed=1, precision=[2, None, 0], scale=[None, None, None])
    np.random.seed(3)
    xopt, fopt = bogp.argmax_restart(bogp.EI(model=gp), sp, eval_budget=30000, optimizer="sweep-device-lhs")
    assert np.round(xopt[0], 2) == xopt[0] and np.round(xopt[2], 0) == xopt[2] and all(abs(v) <= 5 for v in xopt)
    np.testing.assert_allclose(float(np.ravel(bogp.EI(model=gp)(np.array(xopt).reshape(1, -1)))[0]), fopt, rtol=1e-9)


def test_reml_with_polynomial_trends_on_the_device():
    """bogp_nll_restricted with the linear (p = 4) and quadratic (p = 10) bases against the reference's tables (G27): value
    1e-9, gradient 1e-6; then `fit(likelihood="restricted")` with a linear trend runs to a finite optimum whose value the
    oracle reproduces."""
    from conftest import load_golden

    g = load_golden("G27_reml_trend_tables")
    eng = _lib.Engine(0)
    eng.set_train(g["X"], g["y"])
    n = 0
    for tid in (1, 2):
        beta = g["t%d_beta" % tid]
        for kid in (0, 2):
            for mid in (0, 1, 2):
                for tname in ("uk", "sk"):
                    key = "t%d_k%d_m%d_%s" % (tid, kid, mid, tname)
                    for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                        llf, grad = eng.nll_restricted(kid, mid, p, 1e-6 if mid == 1 else 0.0, tname == "uk", beta, eval_grad=True, trend=tid)
                        if np.isneginf(v):
                            assert np.isneginf(llf)
                        else:
                            np.testing.assert_allclose(llf, v, rtol=1e-9, err_msg=key)
                        np.testing.assert_allclose(grad, gr, rtol=1e-6, atol=1e-8 * np.abs(gr).max(), err_msg=key)
                        assert eng.nll_restricted(kid, mid, p, 1e-6 if mid == 1 else 0.0, tname == "uk", beta, trend=tid) == llf
                        n += 1
    assert n == 72
    eng.close()
    X, y = g["X"], g["y"]
    gp = bogp.GaussianProcess(mean=bogp.trend.linear_trend(3), corr="matern", thetaL=[1e-2] * 3, thetaU=[1e2] * 3, nugget=1e-6,
                              likelihood="restricted", random_start=2, eval_budget=80)  # fmt: skip
    np.random.seed(5)
    assert gp.fit(X, y) is gp and gp.is_fitted and np.isfinite(gp.log_likelihood_)
    par = np.concatenate([np.ravel(gp.par[k]) for k in ("theta", "sigma2")])
    ref = O.log_likelihood_restricted(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, noise_var=1e-6, trend=O.TREND_LINEAR, estimate_trend=True)
    np.testing.assert_allclose(gp.log_likelihood_, ref, rtol=1e-9)
    mu, mse = gp.predict(X[:5], eval_MSE=True)
    assert mu.shape == (5, 1) and np.all(mse >= 0)


@pytest.mark.parametrize("fixture", ["G28_driver_trace", "G29_driver_trace_bfgs", "G30_fmin_trace"])
def test_replay_of_the_real_driver_trace(fixture):
    """VERDICT r01 weak 4, joined by data: G28 holds every engine call the unmodified `bayes_optim.ParallelBO` made in the build
    container (through `bogp.install`, on the oracle-backed engine) with its arguments and its answer -- 4 fits (178 likelihood
    evaluations of the host L-BFGS-B loop), commits, state read-backs, candidate uploads, posterior passes, 3 fused top-k sweeps.
    The same calls, in the same order, on the DEVICE engine must give the same answers: then the device-backed run of the real
    driver is the recorded run."""
    import json

    from support.trace_codec import decode

    from conftest import load_golden

    # G29: the plain `BO` with EI + multi-restart L-BFGS-B (one-point posterior / gradient calls); G30: `bayes_optim.fmin()`
    # itself after `bogp.install()` -- the model fmin builds, its 4 fits (613 likelihood evaluations) and its BFGS-driven asks
    g = load_golden(fixture)
    index = json.loads(str(g["index"]))
    assert len(index) == int(g["n_calls"]) >= 200
    eng = _lib.Engine(0)
    seen = {}
    try:
        for n, node in enumerate(index):
            c = decode(node, g)
            name, args, kw, ref, err = c["name"], c["args"], c["kwargs"], c["out"], c["err"]
            seen[name] = seen.get(name, 0) + 1
            where = "call %d: %s" % (n, name)
            if err is not None:
                with pytest.raises(_lib.BogpError):
                    getattr(eng, name)(*args, **kw)
                continue
            out = getattr(eng, name)(*args, **kw)
            if name in ("set_train", "upload_candidates", "select_target"):
                continue
            if name in ("nll", "nll_restricted"):
                if isinstance(ref, (tuple, list)):
                    np.testing.assert_allclose(out[0], ref[0], rtol=1e-6, err_msg=where)
                    np.testing.assert_allclose(np.ravel(out[1]), np.ravel(ref[1]), rtol=1e-6, atol=1e-8 * (1 + np.abs(ref[1]).max()), err_msg=where)
                else:
                    np.testing.assert_allclose(out, ref, rtol=1e-6, err_msg=where)
            elif name == "commit":
                np.testing.assert_allclose(out, ref, rtol=1e-6, err_msg=where)
            elif name == "get_state":
                for key in ("gamma", "rho", "Yt", "Ft", "sigma2", "noise_var", "beta", "G"):
                    if key in ref and ref[key] is not None:
                        scale = max(1.0, float(np.max(np.abs(ref[key]))))
                        np.testing.assert_allclose(np.ravel(out[key]), np.ravel(ref[key]), rtol=1e-6, atol=1e-9 * scale, err_msg=where + " " + key)
                if "C" in ref and ref["C"] is not None and out.get("C") is not None:
                    np.testing.assert_allclose(np.tril(out["C"]), np.tril(ref["C"]), rtol=1e-6, atol=1e-9, err_msg=where + " C")
            elif name == "predict":
                np.testing.assert_allclose(np.ravel(out[0]), np.ravel(ref[0]), rtol=1e-6, atol=1e-9, err_msg=where)
                if ref[1] is not None:
                    np.testing.assert_allclose(np.ravel(out[1]), np.ravel(ref[1]), rtol=1e-6, atol=1e-12, err_msg=where)
            elif name in ("sweep", "sweep_topk"):
                rv, ri = np.asarray(ref[0], float), np.asarray(ref[1])
                ov, oi = np.asarray(out[0], float), np.asarray(out[1])
                np.testing.assert_allclose(ov, rv, rtol=1e-6, atol=1e-300, err_msg=where)
                rv2, ri2, oi2 = np.atleast_2d(rv), np.atleast_2d(ri), np.atleast_2d(oi)
                for row in range(rv2.shape[0]):  # indices exactly wherever the reference's values separate the candidates
                    v = rv2[row]
                    rel = np.abs(np.diff(v)) / np.maximum(np.abs(v[:-1]), 1e-300) if len(v) > 1 else np.array([])
                    firm = np.r_[True, rel > 1e-9] & np.r_[rel > 1e-9, True] if len(v) > 1 else np.array([True])
                    np.testing.assert_array_equal(oi2[row][firm], ri2[row][firm], err_msg=where)
            elif name == "gradient":
                np.testing.assert_allclose(np.ravel(out[0]), np.ravel(ref[0]), rtol=1e-6, atol=1e-9, err_msg=where)
                np.testing.assert_allclose(np.ravel(out[1]), np.ravel(ref[1]), rtol=1e-6, atol=1e-9, err_msg=where)
            else:
                raise AssertionError("unhandled recorded call " + name)
    finally:
        eng.close()
    if fixture == "G28_driver_trace":
        assert seen.get("nll", 0) >= 100 and seen.get("sweep_topk", 0) == 3 and seen.get("commit", 0) == 4
    elif fixture == "G29_driver_trace_bfgs":
        assert seen.get("gradient", 0) >= 20 and seen.get("sweep", 0) >= 20 and seen.get("commit", 0) == 3
    else:
        assert seen.get("nll", 0) >= 500 and seen.get("gradient", 0) >= 40 and seen.get("commit", 0) == 4


@pytest.mark.parametrize("N,d,M", [(2048, 20, 200_000), (700, 7, 300_001), (300, 5, 50_000), (1100, 12, 20)])
def test_lazy_upload_gives_the_sweep_of_the_plain_upload(N, d, M):
    """bogp_candidates_upload_lazy: the rows of chunk c + 1 are copied on a copy stream while chunk c is contracted.  Same winners,
    same posterior, bit for bit, as after the plain upload -- over several chunks (C3-size model), a ragged last chunk, the one-launch
    sweep (N <= 512: the upload is finished first) and the small-batch path; the host array may be dropped once the sweep returned."""
    from bogp import _lib

    rng = np.random.default_rng(N + M)
    X = rng.uniform(-5, 5, size=(N, d))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    eng = _lib.Engine(0)
    try:
        eng.set_train(X, y)
        eng.commit(_lib.KERNEL_MATERN52, _lib.MODE_NOISY, np.r_[np.full(d, 0.2 / d), 0.9], 1e-6, True, 0.0)
        Xs = rng.uniform(-5, 5, size=(M, d))
        acq = [(_lib.ACQ_MGFI, 2.0), (_lib.ACQ_EI, 0.0)]
        pl = float(y.min())
        eng.upload_candidates(Xs)
        want = eng.sweep(acq, pl, True)
        want_k = eng.sweep_topk(acq, pl, True, 8)
        want_mu, want_mse = eng.predict()
        for _ in range(2):
            tmp = Xs.copy()
            eng.upload_candidates(tmp, lazy=True)
            got = eng.sweep(acq, pl, True)
            tmp[:] = np.nan  # the sweep has returned: every row is resident, the host rows are free
            del tmp
            np.testing.assert_array_equal(got[0], want[0])
            np.testing.assert_array_equal(got[1], want[1])
            np.testing.assert_array_equal(eng.read_candidates(got[1]), Xs[got[1]])
            gk = eng.sweep_topk(acq, pl, True, 8)  # a second consumer of the same (now resident) candidates
            np.testing.assert_array_equal(gk[0], want_k[0])
            np.testing.assert_array_equal(gk[1], want_k[1])
        eng.upload_candidates(Xs.copy(), lazy=True)
        mu, mse = eng.predict()
        np.testing.assert_array_equal(mu, want_mu)
        np.testing.assert_array_equal(mse, want_mse)
        # lazy upload straight into read / a new upload before any sweep: nothing dangles
        eng.upload_candidates(Xs.copy(), lazy=True)
        np.testing.assert_array_equal(eng.read_candidates(np.array([0, M - 1])), Xs[[0, M - 1]])
        eng.upload_candidates(Xs[: max(1, M // 2)].copy(), lazy=True)
        eng.upload_candidates(Xs[:7].copy())
        assert eng.sweep(acq, pl, True)[1].max() < 7
    finally:
        eng.close()
