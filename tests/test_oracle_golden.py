"""The oracle (oracle/gp_oracle.py) against golden vectors produced by importing the reference
(oracle/make_golden.py).  CPU only.  Tolerances: 1e-12 relative on values the reference computes with the
same LAPACK/NumPy calls (they are in practice bit-identical); argmax indices exact."""
import numpy as np
import pytest

from conftest import load_golden, state_from_golden
from oracle import gp_oracle as O

RT = 1e-12
STATE_FILES = ["G1_se_sk_noisy", "G2_m32_ok_noisy", "G3_m52_sk_noisy", "G4_se_ok_noiseless", "G5_se_sk_noise_estim", "G7_edges",
               "G12_absexp_ok_noisy", "G13_linear_uk_se", "G14_quadratic_uk_m32", "G15_linear_sk_se"]


def close(a, b, rtol=RT, atol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    np.testing.assert_allclose(a.reshape(b.shape), b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", STATE_FILES)
def test_state_matches_reference(name):
    g = load_golden(name)
    st = state_from_golden(g)
    close(st.llf, g["llf"])
    close(st.C, g["C"])
    close(st.gamma, g["gamma"], rtol=1e-10)
    close(st.rho, g["rho"], rtol=1e-10, atol=1e-13)
    close(st.Yt, g["Yt"], rtol=1e-10, atol=1e-13)
    close(st.sigma2, g["sigma2"])
    close(st.beta, g["beta"], rtol=1e-10)
    if st.estimate_trend:
        close(st.Ft, g["Ft"])
        close(st.G, g["G"])
        close(st.Q, g["Q"])


@pytest.mark.parametrize("name", STATE_FILES)
def test_predict_matches_reference(name):
    g = load_golden(name)
    st = state_from_golden(g)
    mu, mse = O.predict(st, g["Xs"])
    close(mu, g["mu"], rtol=1e-10, atol=1e-13)
    close(mse, g["mse"], rtol=1e-9, atol=1e-13)
    # chunked evaluation is the same function row by row
    mu2, mse2 = O.predict_chunked(st, g["Xs"], chunk=37)
    close(mu2, g["mu"], rtol=1e-10, atol=1e-13)
    close(mse2, g["mse"], rtol=1e-9, atol=1e-13)


def _acq_cases(g, prefix=""):
    for k in g:
        if not k.startswith(prefix) or k.startswith(prefix + "argmax_") or k.startswith("dx_"):
            continue
        base = k[len(prefix):]
        if base == "EI":
            yield k, O.ACQ_EI, 0.0
        elif base.startswith("EpsilonPI_"):
            yield k, O.ACQ_EPSILON_PI, float(base.split("_")[1])
        elif base.startswith("UCB_"):
            yield k, O.ACQ_UCB, float(base.split("_")[1])
        elif base.startswith("MGFI_"):
            yield k, O.ACQ_MGFI, float(base.split("_")[1])


@pytest.mark.parametrize("name", STATE_FILES)
def test_acquisitions_match_reference_row_by_row(name):
    g = load_golden(name)
    st = state_from_golden(g)
    # acquisition values from the REFERENCE's stored (mu, mse): isolates the closed forms + guards
    mu, mse = g["mu"][:, 0], g["mse"][:, 0]
    variants = [("", True, None)]
    if name == "G7_edges":
        variants += [("max_", False, None), ("plg_", True, -0.3)]
    for prefix, minimize, plugin in variants:
        pl = O.plugin_value(st.y, minimize, plugin)
        close(pl, g[prefix + "plugin_eff"])
        n = 0
        for key, acq, par in _acq_cases(g, prefix):
            if prefix == "" and (key.startswith("max_") or key.startswith("plg_")):
                continue
            v = O.acquisition(acq, par, mu, mse, pl, st.sigma2[0], minimize)
            ref = g[key]
            # the stored rows come from single-row predict() calls, the stored (mu, mse) from one batched
            # call: BLAS rounds the two differently in the last bits, and deep-tail EI/MGFI amplify that by
            # ~z^3 (cancellation), hence 1e-9 rather than 1e-12 here
            ok = np.ones(len(v), bool)
            if acq in (O.ACQ_EPSILON_PI, O.ACQ_MGFI):
                # EpsilonPI has no small-variance guard (acquisition_fun.py:208-217) and MGFI's guard is
                # sd <= 1e-8 (:274): on rows whose MSE is rounding noise around 0 (a training point in a
                # noiseless model) the value is Phi(noise/noise) -- not a reproducible number even between two
                # calls of the reference; only its range is checked there
                ok = mse > 1e-12 * st.sigma2[0]
                assert np.all((v[~ok] >= 0) | np.isnan(v[~ok]))
            np.testing.assert_allclose(v[ok], ref[ok], rtol=1e-9, atol=1e-300, equal_nan=True, err_msg=key)
            if not ok.all():
                continue
            assert O.nan_first_argmax(v) == int(g[prefix + "argmax_" + key[len(prefix):]][0]), key
            n += 1
        assert n >= 2


@pytest.mark.parametrize("name", STATE_FILES)
def test_sweep_argmax_matches_reference(name):
    """End to end through the oracle's own posterior: same argmax index as the reference row loop."""
    g = load_golden(name)
    st = state_from_golden(g)
    acq = [(a, p) for _, a, p in _acq_cases(g, "") if True]
    keys = [k for k, _, _ in _acq_cases(g, "")]
    keep = [i for i, k in enumerate(keys) if not (k.startswith("max_") or k.startswith("plg_"))]
    acq, keys = [acq[i] for i in keep], [keys[i] for i in keep]
    best, idx = O.sweep(st, g["Xs"], acq, chunk=64)
    for k, b, i in zip(keys, best, idx):
        assert i == int(g["argmax_" + k][0]), k
        np.testing.assert_allclose(b, g[k][i], rtol=1e-8, atol=1e-300)


def test_edge_semantics():
    """candidate == training point in a noiseless model -> MSE exactly 0 -> EI 0,
    MGFI 0, EpsilonPI in {0,1}, UCB = mu; t clamps at 22.36."""
    g = load_golden("G7_edges")
    assert np.all(g["mse"][:6, 0] <= 1e-15) and np.any(g["mse"][:6, 0] == 0.0)  # clipped to 0 or rounding noise
    assert np.all(g["EI"][:6] == 0.0) and np.all(g["MGFI_1"][:6] == 0.0)
    assert set(np.unique(g["EpsilonPI_1e-10"][:6])) <= {0.0, 1.0}
    np.testing.assert_allclose(g["UCB_0.5"][:6], g["mu"][:6, 0], atol=2e-8)
    np.testing.assert_allclose(
        g["MGFI_100"], O.mgfi(g["mu"][:, 0], g["mse"][:, 0], float(g["plugin_eff"][0]), 22.36), rtol=1e-11
    )  # MGFI(t=100).t == 22.36 (clamp, acquisition_fun.py:260-263)


@pytest.mark.parametrize("name", ["G1_se_sk_noisy", "G2_m32_ok_noisy", "G4_se_ok_noiseless", "G12_absexp_ok_noisy",
                                  "G13_linear_uk_se", "G15_linear_sk_se"])
def test_gradient_matches_reference(name):
    g = load_golden(name)
    st = state_from_golden(g)
    for i in range(len(g["grad_mu"])):
        dmu, dmse = O.gradient(st, g["Xs"][i])
        close(dmu, g["grad_mu"][i], rtol=1e-9, atol=1e-13)
        close(dmse, g["grad_mse"][i], rtol=1e-8, atol=1e-13)


def test_llf_tables():
    g = load_golden("G6_llf_tables")
    X, y = g["X"], g["y"]
    n = 0
    for kid in (0, 2):
        for mid in (0, 1, 2):
            for tname in ("sk", "ok"):
                key = "k%d_m%d_%s" % (kid, mid, tname)
                for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                    out = O.log_likelihood_concentrated(
                        p, X, y, kid, mid, noise_var=1e-6 if mid == 1 else 0.0,
                        estimate_trend=(tname == "ok"), beta=0.0, eval_grad=True)  # fmt: skip
                    if np.isneginf(v):
                        assert np.isneginf(out[0])
                        continue
                    close(out[0], v, rtol=1e-11)
                    close(out[1], gr, rtol=1e-8, atol=1e-9)
                    n += 1
    assert n >= 30


def test_mid_size():
    g = load_golden("G8_mid")
    rng = np.random.default_rng(8)
    X = rng.uniform(-5, 5, size=(512, 10))
    y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std()).reshape(-1, 1)
    st = O.make_state(g["par"], X, y, O.KERNEL_SE, O.MODE_NOISY, noise_var=1e-6)
    close(st.llf, g["llf"], rtol=1e-11)
    close(st.gamma, g["gamma"], rtol=1e-7, atol=1e-9)
    mu, mse = O.predict_chunked(st, g["Xs"], 512)
    close(mu, g["mu"], rtol=1e-8, atol=1e-11)
    close(mse, g["mse"], rtol=1e-8, atol=1e-12)
    v = O.ei(mu[:, 0], mse[:, 0], O.plugin_value(y, True), st.sigma2[0])
    np.testing.assert_allclose(v, g["EI"], rtol=1e-7, atol=1e-300)
    assert int(np.argmax(v)) == int(g["argmax_EI"][0])


def test_fmin_plumbing_invariants_recorded():
    g = load_golden("G9_fmin_plumbing")
    assert int(g["n_ret"]) == 5 and int(g["n_x"]) == 2 and int(g["n_iter"]) == 21 and int(g["n_eval"]) == 30


def test_philox_known_answers():
    """Known-answer vectors of Philox4x32-10 from the Random123 distribution (kat_vectors): they pin the generator
    that the device kernel and oracle/philox.py both implement."""
    from oracle import philox as P

    def run(c, k):
        out = P.philox4x32_10(*[np.array([x], dtype=np.uint32) for x in c], np.uint32(k[0]), np.uint32(k[1]))
        return [int(o[0]) for o in out]

    assert run((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF)) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_philox_uniform_box_properties():
    from oracle import philox as P

    lo, hi = np.array([-5.0, 0.0, 2.0]), np.array([5.0, 1.0, 2.5])
    X = P.uniform_box(lo, hi, 4001, seed=12345)
    assert X.shape == (4001, 3) and np.all(X >= lo) and np.all(X < hi + 1e-15)
    # shards of one stream are independent of how it is cut
    parts = np.vstack([P.uniform_box(lo, hi, 1000, 12345, 0), P.uniform_box(lo, hi, 3001, 12345, 1000)])
    np.testing.assert_array_equal(parts, X)
    assert abs(X[:, 0].mean()) < 0.2 and abs(X[:, 1].mean() - 0.5) < 0.02
    assert not np.array_equal(P.uniform_box(lo, hi, 10, 1), P.uniform_box(lo, hi, 10, 2))


def test_latin_hypercube_restatement_properties():
    """One point per stratum and dimension (what pyDOE's classic design guarantees, search_space.py:747-751), for
    every n incl. non-powers of two; shards of one design are independent of how it is cut."""
    from oracle import philox as P

    lo, hi = np.array([-5.0, 0.0, 2.0]), np.array([5.0, 1.0, 2.5])
    for n in (1, 2, 3, 7, 64, 1000, 4097):
        X = P.lhs_box(lo, hi, n, seed=99)
        strata = np.floor((X - lo) / (hi - lo) * n).astype(int)
        for k in range(3):
            assert sorted(strata[:, k].tolist()) == list(range(n)), (n, k)
    whole = P.lhs_box(lo, hi, 1000, 7)
    parts = np.vstack([P.lhs_box(lo, hi, 400, 7, 0, 1000), P.lhs_box(lo, hi, 600, 7, 400, 1000)])
    np.testing.assert_array_equal(parts, whole)
    # per-dimension permutations differ from each other and between seeds
    big = P.lhs_box(lo, hi, 4096, 5)
    assert np.abs(np.corrcoef(big.T) - np.eye(3)).max() < 0.08
    assert not np.array_equal(P.lhs_box(lo, hi, 64, 1), P.lhs_box(lo, hi, 64, 2))
    # the keyed permutation is a bijection on awkward domain sizes
    for n in (2, 5, 17, 1025):
        pi = P.lhs_permutation(np.arange(n), np.zeros(n, dtype=np.uint32), 3, n)
        assert sorted(pi.tolist()) == list(range(n))


def test_maximin_restatement_follows_pydoe():
    """pyDOE's _lhsmaximin (what the reference's "LHS" asks for, search_space.py:751): of `iterations` hypercubes keep the
    FIRST with the largest scipy pdist minimum.  The restatement's distance equals scipy's pdist bit for bit."""
    from scipy.spatial.distance import pdist

    from oracle import philox as P

    rng = np.random.default_rng(3)
    for n, d in ((2, 1), (5, 3), (64, 7), (257, 20)):
        X = rng.random((n, d))
        assert P.min_pdist(X) == float(pdist(X).min()), (n, d)
    lo, hi = np.array([-5.0, 0.0, 2.0]), np.array([5.0, 1.0, 2.5])
    X, dist, t = P.lhs_maximin_box(lo, hi, 40, seed=11, iterations=5)
    trials = [P.lhs_box(np.zeros(3), np.ones(3), 40, (11 + 0x9E3779B97F4A7C15 * i) & (2**64 - 1)) for i in range(5)]
    dists = [float(pdist(T).min()) for T in trials]
    assert t == int(np.argmax(dists)) and dist == max(dists)  # np.argmax: first maximum, like `if maxdist < min(d)`
    np.testing.assert_array_equal(X, lo + (hi - lo) * trials[t])
    assert dist >= dists[0]  # never worse than the plain design of the same stream


def test_sobol_restatement_equals_the_reference_generator():
    """RealSpace._sample(method="sobol") (search_space.py:752-753) = (ub - lb) * i4_sobol_generate(dim, N) + lb; in
    this image `sobol_seq` resolves to scipy's unscrambled generator minus its first point."""
    import warnings

    from scipy.stats import qmc

    from oracle import philox as P

    for d in (1, 3, 20, 50):
        lo, hi = np.linspace(-5, -1, d), np.linspace(1, 5, d)
        sv = qmc.Sobol(d=d, scramble=False)._sv
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = (hi - lo) * qmc.Sobol(d=d, scramble=False).random(1000 + 1)[1:] + lo
        np.testing.assert_array_equal(P.sobol_box(lo, hi, 1000, sv, 1), ref)
        np.testing.assert_array_equal(P.sobol_box(lo, hi, 300, sv, 701), ref[700:])


def test_absexp_llf_tables():
    g = load_golden("G12_absexp_ok_noisy")
    n = 0
    for mid in (0, 1, 2):
        for tname in ("sk", "ok"):
            key = "t_m%d_%s" % (mid, tname)
            for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                out = O.log_likelihood_concentrated(p, g["X"], g["y"], O.KERNEL_ABSEXP, mid, noise_var=1e-6 if mid == 1 else 0.0,
                                                    estimate_trend=(tname == "ok"), beta=0.0, eval_grad=True)  # fmt: skip
                if np.isneginf(v):
                    assert np.isneginf(out[0])
                    continue
                close(out[0], v, rtol=1e-11)
                close(out[1], gr, rtol=1e-8, atol=1e-9)
                n += 1
    assert n >= 12


@pytest.mark.parametrize("name", ["G13_linear_uk_se", "G14_quadratic_uk_m32", "G15_linear_sk_se"])
def test_trend_llf_tables(name):
    """Polynomial trends with p > 1 columns (trend.py:94-142): likelihood + gradient in the three modes."""
    g = load_golden(name)
    est = bool(g["estimate_trend"])
    beta = None if est else np.asarray(g["beta"], float).ravel()
    n = 0
    for mid in (0, 1, 2):
        key = "t_m%d" % mid
        for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
            out = O.log_likelihood_concentrated(p, g["X"], g["y"], int(g["kernel"]), mid, noise_var=1e-6 if mid == 1 else 0.0,
                                                trend=int(g["trend"]), estimate_trend=est, beta=beta, eval_grad=True)  # fmt: skip
            if np.isneginf(v):
                assert np.isneginf(out[0])
                continue
            close(out[0], v, rtol=1e-11)
            close(out[1], gr, rtol=1e-8, atol=1e-9)
            n += 1
    assert n >= 9


def test_quadratic_trend_has_no_jacobian():
    g = load_golden("G14_quadratic_uk_m32")
    with pytest.raises(NotImplementedError):
        O.gradient(state_from_golden(g), g["Xs"][0])


def test_reml_tables():
    """The restricted likelihood (gpr.py:813-918): value + gradient, three modes x {sk, ok} x {SE, Matern-3/2}."""
    g = load_golden("G16_reml_tables")
    n = 0
    for kid in (0, 2):
        for mid in (0, 1, 2):
            for tname in ("sk", "ok"):
                key = "k%d_m%d_%s" % (kid, mid, tname)
                for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                    out = O.log_likelihood_restricted(p, g["X"], g["y"], kid, mid, noise_var=1e-6 if mid == 1 else 0.0,
                                                      estimate_trend=(tname == "ok"), beta=0.0, eval_grad=True)  # fmt: skip
                    if np.isneginf(v):
                        assert np.isneginf(out[0])
                    else:
                        close(out[0], v, rtol=1e-11)
                    close(out[1], gr, rtol=1e-8, atol=1e-9)
                    n += 1
    assert n == 48


def test_reml_isotropic_theta_tables():
    """G38: the oracle's restricted likelihood reproduces the reference's slice-indexed gradient for an isotropic theta (d = 1 .. 4)."""
    g = load_golden("G38_reml_isotropic")
    n = 0
    for d in (1, 2, 3, 4):
        for kid in (0, 2):
            for mid in (0, 1, 2):
                for tname in ("sk", "ok"):
                    key = "d%d_k%d_m%d_%s" % (d, kid, mid, tname)
                    for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                        ov, og = O.log_likelihood_restricted(p, g["X%d" % d], g["y%d" % d], kid, mid, 1e-6 if mid == 1 else 0.0,
                                                             estimate_trend=tname == "ok", beta=0.0, eval_grad=True)  # fmt: skip
                        close(ov, v, rtol=1e-12)
                        close(np.ravel(og), gr, rtol=1e-9, atol=1e-10 * np.abs(gr).max())
                        n += 1
    assert n == 144


def test_reml_multitarget_tables():
    """The restricted likelihood on y with 2 / 3 columns (G33, generated by importing the reference): the value the reference's
    arithmetic yields -- scalar terms broadcast over the n_t x n_t matrix rho^T rho, everything summed (gpr.py:861-866)."""
    g = load_golden("G33_reml_multitarget")
    n = 0
    for T in (2, 3):
        for kid in (0, 2):
            for mid in (0, 1, 2):
                key = "T%d_k%d_m%d" % (T, kid, mid)
                for p, v in zip(g[key + "_par"], g[key + "_llf"]):
                    out = O.log_likelihood_restricted(p, g["X"], g["Y"][:, :T], kid, mid, noise_var=1e-4 if mid == 1 else 0.0,
                                                      estimate_trend=False, beta=float(g["beta"]))  # fmt: skip
                    if np.isneginf(v):
                        assert np.isneginf(out)
                    else:
                        close(out, v, rtol=1e-11)
                    n += 1
    assert n == 36


MT_NV = {0: 0.0, 1: 1e-3, 2: 0.0}  # nugget of G17's noisy-mode model


def test_multitarget_tables_and_states():
    """y with three columns (gpr.py:463, 490, 502-505, 931-1040): the likelihood is summed over targets, its gradient
    sums gamma gamma^T over ALL targets in the theta rows; sigma2 / gamma / mu / MSE get one column per target."""
    g = load_golden("G17_multitarget")
    X, Y, Xs = g["X"], g["y"], g["Xs"]
    assert Y.shape[1] == 3
    n = 0
    for kid in (0, 2):
        for mid in (0, 1, 2):
            key = "k%d_m%d" % (kid, mid)
            for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                out = O.log_likelihood_concentrated(p, X, Y, kid, mid, noise_var=MT_NV[mid], beta=0.0, eval_grad=True)
                close(out[0], v, rtol=1e-11)
                close(out[1], gr, rtol=1e-8, atol=1e-9)
                n += 1
            st = O.make_state(g[key + "_par"][0], X, Y, kid, mid, noise_var=MT_NV[mid], beta=0.0)
            close(st.sigma2, g[key + "_st_sigma2"], rtol=1e-11)
            close(st.gamma, g[key + "_st_gamma"], rtol=1e-7, atol=1e-9)
            close(st.rho, g[key + "_st_rho"], rtol=1e-8, atol=1e-11)
            mu, mse = O.predict(st, Xs)
            assert mu.shape == (len(Xs), 3) and mse.shape == (len(Xs), 3)
            close(mu, g[key + "_mu"], rtol=1e-8, atol=1e-10)
            close(mse, g[key + "_mse"], rtol=1e-7, atol=1e-10)
    assert n == 24


def test_isotropic_theta_tables():
    """One theta for d = 3 dimensions: the value is the ordinary likelihood, the gradient is the reference's
    parameter-indexed slice of the per-dimension tensor (gpr.py:1001-1037)."""
    g = load_golden("G18_isotropic_tables")
    n = 0
    for kid in (0, 2, 4):
        for mid in (0, 1, 2):
            for tname in ("sk", "ok"):
                key = "k%d_m%d_%s" % (kid, mid, tname)
                for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                    out = O.log_likelihood_concentrated(p, g["X"], g["y"], kid, mid, noise_var=1e-6 if mid == 1 else 0.0,
                                                        estimate_trend=(tname == "ok"), beta=0.0, eval_grad=True)  # fmt: skip
                    if np.isneginf(v):
                        assert np.isneginf(out[0])
                        continue
                    close(out[0], v, rtol=1e-11)
                    close(out[1], gr, rtol=1e-8, atol=1e-9)
                    n += 1
    assert n >= 40


def test_hessian_and_prior_cov_restatements():
    """GaussianProcess.Hessian (gpr.py:578-598) and prior_cov (gpr.py:318-353) against the reference's outputs (G19)."""
    g = load_golden("G19_hessian_prior_cov")
    for tag, mode, est in (("sk", O.MODE_NOISY, False), ("ok", O.MODE_NOISELESS, True)):
        st = O.make_state(g[tag + "_par"], g[tag + "_X"], g[tag + "_y"], O.KERNEL_SE, mode, 1e-6 if mode == O.MODE_NOISY else 0.0,
                          estimate_trend=est, beta=None if est else 0.0)  # fmt: skip
        for p, H in zip(g[tag + "_P"], g[tag + "_H"]):
            close(O.hessian(st, p), H, rtol=1e-9, atol=1e-12)
        close(O.prior_cov(st, g[tag + "_P"], corr_only=True), g[tag + "_corr"], rtol=1e-13, atol=0)
        close(O.prior_cov(st, g[tag + "_P"]), g[tag + "_cov"], rtol=1e-9, atol=0)
    with pytest.raises(NotImplementedError):
        O.hessian(state_from_golden(load_golden("G2_m32_ok_noisy")), np.zeros(2))


@pytest.mark.parametrize("name", ["G25_cubic_ok_noisy", "G26_genexp_sk_noisy", "G31_matern_nu08_ok_noisy", "G32_matern_nu37_sk_noisy"])
def test_value_only_kernels_against_the_reference(name):
    """cubic and generalized_exponential (round 2), the general-nu Matern arm (round 4; nu = 0.8 / 3.7 through scipy.special.kv): state, posterior, criteria row by row, argmax and the likelihood VALUE
    tables of the three modes (several entries are -inf: the llf > 0 rejection, gpr.py:981-982) -- the oracle against the
    reference's own outputs."""
    g = load_golden(name)
    st = state_from_golden(g)
    np.testing.assert_allclose(st.llf, float(g["llf"]), rtol=1e-12)
    np.testing.assert_allclose(st.C, g["C"], rtol=0, atol=1e-12 * np.abs(g["C"]).max())
    np.testing.assert_allclose(st.gamma, g["gamma"], rtol=1e-9, atol=1e-12 * np.abs(g["gamma"]).max())
    mu, mse = O.predict(st, g["Xs"])
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(mse, g["mse"], rtol=1e-9, atol=1e-14)
    pl = O.plugin_value(st.y, True)
    solid = g["mse"][:, 0] > 1e-9 * st.sigma2[0]
    for key, a, p in (("EI", O.ACQ_EI, 0.0), ("EpsilonPI_1e-10", O.ACQ_EPSILON_PI, 1e-10), ("UCB_0.5", O.ACQ_UCB, 0.5),
                      ("MGFI_1", O.ACQ_MGFI, 1.0), ("MGFI_2", O.ACQ_MGFI, 2.0), ("MGFI_100", O.ACQ_MGFI, 100.0)):  # fmt: skip
        v = O.acquisition(a, p, mu[:, 0], mse[:, 0], pl, st.sigma2[0], True)
        np.testing.assert_allclose(v[solid], g[key][solid], rtol=1e-9, atol=1e-300)
        if int(g["argmax_" + key][0]) != 5:  # (row 5 sits on a training point)
            assert int(np.argmax(v)) == int(g["argmax_" + key][0])
    kid = int(g["kernel"])
    for mid in (0, 1, 2):
        for tname, est in (("sk", False), ("ok", True)):
            P, L = g["t_m%d_%s_par" % (mid, tname)], g["t_m%d_%s_llf" % (mid, tname)]
            for p_, l_ in zip(P, L):
                out = O.log_likelihood_concentrated(p_, g["X"], g["y"], kid, mid, noise_var=1e-6 if mid == 1 else 0.0,
                                                    estimate_trend=est, beta=0.0)  # fmt: skip
                if np.isfinite(l_):
                    np.testing.assert_allclose(out, l_, rtol=1e-11)
                else:
                    assert out == l_


def test_reml_with_polynomial_trends_against_the_reference():
    """G27: log_likelihood_restricted (gpr.py:813-918) with linear / quadratic bases (p > 1), value and gradient."""
    g = load_golden("G27_reml_trend_tables")
    n = 0
    for tid in (1, 2):
        for kid in (0, 2):
            for mid in (0, 1, 2):
                for tname in ("uk", "sk"):
                    key = "t%d_k%d_m%d_%s" % (tid, kid, mid, tname)
                    for p, v, gr in zip(g[key + "_par"], g[key + "_llf"], g[key + "_grad"]):
                        out = O.log_likelihood_restricted(p, g["X"], g["y"], kid, mid, noise_var=1e-6 if mid == 1 else 0.0, trend=tid,
                                                          estimate_trend=(tname == "uk"), beta=g["t%d_beta" % tid], eval_grad=True)  # fmt: skip
                        if np.isneginf(v):
                            assert np.isneginf(out[0])
                        else:
                            close(out[0], v, rtol=1e-10)
                        close(out[1], gr, rtol=1e-7, atol=1e-8)
                        n += 1
    assert n == 72


def test_device_bessel_algorithm_restated_on_the_cpu_against_the_truth_table():
    """K_nu(x) as csrc/bogp_device.h computes it since r05 (Chebyshev expansions of gam1 / gam2, Temme's series for x <= 1 with the powers
    from pow(), the trapezoidal rule on the integral representation above, upward recurrence in double-double arithmetic) restated step
    for step in Python and held to the TRUE values of tests/golden/G36_kv_table.npz (mpmath, 40 digits) on every 20th of its 1e5 pairs:
    <= 6 eps, where scipy.special.kv -- what the reference calls (kernel.py:207) -- is up to hundreds of eps off."""
    from oracle.make_kv_table import pairs
    from support.bessel import knu

    g = load_golden("G36_kv_table")
    nu, x = pairs()
    assert float(np.sum(nu) + np.sum(x)) == float(g["nu_x_checksum"])
    true, rlo, sc = g["kv_true"], g["kv_true_rlo"].astype(np.float64), g["kv_scipy"]
    eps = 2.0**-52
    worst = worst_scipy = 0.0
    for i in range(0, len(nu), 20):
        if not (1e-290 < true[i] < 1e290):
            continue
        worst = max(worst, abs((knu(float(nu[i]), float(x[i])) - true[i]) / true[i] - rlo[i]) / eps)
        worst_scipy = max(worst_scipy, abs((sc[i] - true[i]) / true[i] - rlo[i]) / eps)
    assert worst <= 6.0, worst
    assert worst_scipy > 10.0 * worst  # scipy.special.kv itself on the same pairs
