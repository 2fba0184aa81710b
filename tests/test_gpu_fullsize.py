"""BASELINE.json's configurations at their FULL per-GPU sizes against the reference's own answer (top item).

`oracle/make_fullsize_golden.py` pushed every one of the M seeded candidates of C2 / C3 / C4 / C5 through the imported
reference's `GaussianProcess.predict` in the build container and stored, per criterion, the 16 best candidates (index,
value, mu, MSE), the posterior on a fixed 4096-row slice, and sums over all M rows (tests/golden/G20..G23).  Here the
device sweeps the same candidates: the argmax INDEX must equal the reference's, the top-16 set too, values to 1e-6.
Tolerances as in tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import gp_oracle as O
from support.workloads import FULL_SIZE, full_size_problem

pytestmark = pytest.mark.gpu

from bogp import _lib  # noqa: E402

FILES = {"C2": "G20_c2_full", "C3": "G21_c3_full", "C4": "G22_c4_full", "C5": "G23_c5_full"}


@pytest.fixture(scope="module")
def eng():
    e = _lib.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4", "C5"])
def test_full_size_argmax_matches_fixture(eng, cfg):
    if not os.path.exists(os.path.join(GOLDEN, FILES[cfg] + ".npz")):
        pytest.fail("fixture %s.npz is missing: run oracle/make_fullsize_golden.py %s in the build container" % (FILES[cfg], cfg))
    g = load_golden(FILES[cfg])
    w = FULL_SIZE[cfg]
    X, y, par, Xs = full_size_problem(cfg)
    M = w["M"]
    assert float(np.sum(Xs[::997])) == float(g["x_checksum"]) and int(g["M"]) == M  # the very candidates the reference saw
    np.testing.assert_array_equal(np.array(w["acq"], float), g["acq"])
    eng.set_train(X, y)
    llf = eng.commit(w["kernel"], O.MODE_NOISY, par, 1e-6, False, 0.0)
    np.testing.assert_allclose(llf, float(g["llf"]), rtol=1e-9)
    eng.upload_candidates(Xs)
    pl = float(g["plugin"])
    assert pl == O.plugin_value(y, True)
    q = len(w["acq"])
    best, idx = eng.sweep(w["acq"], pl, True)
    # (1) the headline claim: argmax index bit-exact against the reference at full M
    np.testing.assert_array_equal(idx, g["top_idx"][:, 0])
    np.testing.assert_allclose(best, g["top_val"][:, 0], rtol=1e-6)
    np.testing.assert_allclose(best, g["ref_rowwise"][:, 0], rtol=1e-6)  # the reference's own class on that row
    # (2) the 16 best candidates per criterion: same set, same order wherever the reference's values differ by > 1e-9
    tv, ti = eng.sweep_topk(w["acq"], pl, True, 16)
    for c in range(q):
        assert set(ti[c].tolist()) == set(g["top_idx"][c].tolist()), (cfg, c)
        np.testing.assert_allclose(tv[c], g["top_val"][c], rtol=1e-6)
        rel = np.abs(np.diff(g["top_val"][c])) / np.abs(g["top_val"][c][:-1])
        firm = np.r_[True, rel > 1e-9] & np.r_[rel > 1e-9, True]
        np.testing.assert_array_equal(ti[c][firm], g["top_idx"][c][firm])
    # (3) posterior of all M rows: the slice and the top rows row by row, the rest through the sums
    mu, mse = eng.predict()
    s2 = float(g["sigma2"])
    rows = g["slice_rows"]
    np.testing.assert_allclose(mu[rows], g["slice_mu"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse[rows], g["slice_mse"], rtol=1e-6, atol=1e-12 * s2)
    np.testing.assert_allclose(mu[g["top_idx"]], g["top_mu"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mse[g["top_idx"]], g["top_mse"], rtol=1e-6, atol=1e-12 * s2)
    np.testing.assert_allclose(np.sum(mu), float(g["sum_mu"]), rtol=1e-9, atol=1e-9 * M)
    np.testing.assert_allclose(np.sum(mse), float(g["sum_mse"]), rtol=1e-9)
    # (4) how many candidates have a positive criterion value (the plateau of exact zeros decides ties at scale)
    _, _, vals = eng.sweep(w["acq"], pl, True, return_values=True)
    for c in range(q):
        assert abs(int(np.count_nonzero(vals[c] > 0)) - int(g["count_pos"][c])) <= max(2, M // 100000), (cfg, c)
        assert int(np.argmax(vals[c])) == int(g["top_idx"][c, 0])


def test_fit_ensemble_matches_the_reference_distribution():
    """SURVEY row a18 with evidence instead of one anecdote: 54 complete fits by the imported reference (G24) against the
    same 54 fits on the device (same data, same constructor keywords, same np.random seed, the reference's own restart
    loop).  L-BFGS-B is handed the reference's inconsistent gradient (d/d par for a function of log10 par), so last-bit
    differences of the likelihood change which local optimum a restart ends in -- G24's NULL runs (the oracle itself
    with a deterministic 1e-13 perturbation) quantify that.  Asserted:
      * at the REFERENCE's fitted parameters the device likelihood is the reference's (1e-9), every case;
      * at the device's own fitted parameters the device likelihood is the oracle's (1e-9), every case;
      * the paired difference of the final log-likelihood is centred: |median| <= 1e-6, and among the pairs that differ
        by more than 1e-6 neither side wins more than 60 % + the binomial slack of that many pairs;
      * the device does not disagree with the reference more often than the null runs do (+ slack)."""
    import bogp

    g = load_golden("G24_fit_ensemble")
    n = int(g["n_cases"])
    modes = ("noiseless", "noisy", "noise_estim")
    dl, agree_llf = [], 0
    for i in range(n):
        k = "c%02d_" % i
        X, y, d = g[k + "X"], g[k + "y"], int(g["d"][i])
        mode = modes[int(g["mode"][i])]
        gp = bogp.GaussianProcess(mean=bogp.trend.constant_trend(d) if bool(g["ok"][i]) else None,
                                  corr="matern" if bool(g["corr"][i]) else "squared_exponential", thetaL=[1e-3] * d, thetaU=[1e2] * d,
                                  nugget=0 if mode == "noiseless" else 1e-6, noise_estim=mode == "noise_estim", optimizer="BFGS",
                                  wait_iter=3, random_start=5, eval_budget=100 * d)  # fmt: skip
        np.random.seed(int(g["fit_seed"][i]))
        gp.fit(X, y)
        assert gp.is_fitted and modes.index(gp.estimation_mode) == int(g[k + "final_mode"])
        fm = gp.estimation_mode
        kid = O.KERNEL_MATERN32 if bool(g["corr"][i]) else O.KERNEL_SE
        mid = {"noiseless": O.MODE_NOISELESS, "noisy": O.MODE_NOISY, "noise_estim": O.MODE_NOISE_ESTIM}[fm]
        est = bool(g["ok"][i])
        nv = float(np.ravel(gp.noise_var)[0]) if fm == "noisy" else 0.0
        # the device's optimum, judged by the oracle
        own = gp._committed_par
        st = O.make_state(own, X, y, kid, mid, nv, estimate_trend=est, beta=0.0)
        np.testing.assert_allclose(gp.log_likelihood_, st.llf, rtol=1e-9, err_msg="case %d" % i)
        # the reference's optimum, judged by the device
        s2, rnv = float(g[k + "sigma2"][0]), float(g[k + "noise_var"][0])
        ref_par = {"noiseless": g[k + "theta"], "noisy": np.r_[g[k + "theta"], s2],
                   "noise_estim": np.r_[g[k + "theta"], s2 / (s2 + rnv) if s2 + rnv > 0 else 0.5]}[fm]  # fmt: skip
        np.testing.assert_allclose(gp.log_likelihood_concentrated(ref_par), float(g["ref_llf"][i]), rtol=1e-9, err_msg="case %d" % i)
        dl.append(gp.log_likelihood_ - float(g["ref_llf"][i]))
    dl = np.array(dl)
    dn = g["null_llf"] - g["ref_llf"][:, None]
    differ = np.abs(dl) > 1e-6
    better, worse = int(np.sum(dl > 1e-6)), int(np.sum(dl < -1e-6))
    nd = int(differ.sum())
    print("fit ensemble: %d of %d device fits differ from the reference by > 1e-6 (null: %.0f %%); device better %d, worse %d; "
          "median %.3g; null better %d, worse %d" % (nd, n, 100.0 * np.mean(np.abs(dn) > 1e-6), better, worse, np.median(dl),
                                                     int(np.sum(dn > 1e-6)), int(np.sum(dn < -1e-6))))  # fmt: skip
    assert abs(np.median(dl)) <= 1e-6
    slack = 1.5 * np.sqrt(max(nd, 1))  # ~3 sigma of a fair coin over nd pairs
    assert max(better, worse) <= 0.6 * nd + slack, (better, worse)
    null_rate = float(np.mean(np.abs(dn) > 1e-6))
    assert nd <= null_rate * n + 3.0 * np.sqrt(n * max(null_rate * (1 - null_rate), 0.05)) + 1, (nd, null_rate)
