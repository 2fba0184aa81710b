"""Polynomial-trend and restricted-likelihood evaluations on the 128-tile fit path at sizes it did not serve before r06 (3072 < N <= 6080): against the
64-block kernels (BOGP_NO_BIG_FIT=1) and the NumPy oracle; plus a trend sweep (k_mm128's MM_GEN products on the chunk) against the oracle's posterior."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bogp import _lib
from oracle import gp_oracle as O

bad = 0
for N, d, trend in [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or ((3300, 3, 1), (3585, 2, 2), (4100, 4, 1)):
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1) + X[:, 0]
    y = ((y - y.mean()) / y.std() + 0.2 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.3), 0.8]
    out = {}
    for tag, flag in (("big", None), ("small", "1")):
        os.environ.pop("BOGP_NO_BIG_FIT", None)
        if flag: os.environ["BOGP_NO_BIG_FIT"] = flag
        eng = _lib.Engine(0); eng.set_train(X, y)
        llf, grad = eng.nll(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=trend)
        rl, rg = eng.nll_restricted(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True, trend=trend)
        eng.commit(_lib.KERNEL_MATERN32, _lib.MODE_NOISY, par, 1e-6, True, 0.0, trend=trend)
        Xs = rng.uniform(-5, 5, size=(777, d)) if tag == "big" else Xs
        eng.upload_candidates(Xs); mu, mse = eng.predict()
        out[tag] = (llf, grad, rl, rg, mu, mse); eng.close()
    os.environ.pop("BOGP_NO_BIG_FIT", None)
    b, s = out["big"], out["small"]
    ollf = float(O.log_likelihood_concentrated(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True))
    orl = float(O.log_likelihood_restricted(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True))
    st = O.make_state(par, X, y, O.KERNEL_MATERN32, O.MODE_NOISY, 1e-6, trend=trend, estimate_trend=True)
    omu, omse = (np.asarray(v).ravel() for v in O.predict(st, Xs))
    errs = dict(llf_vs_small=abs(b[0] - s[0]) / abs(s[0]), grad_vs_small=np.abs(b[1] - s[1]).max() / np.abs(s[1]).max(), reml_vs_small=abs(b[2] - s[2]) / abs(s[2]),
                llf_vs_oracle=abs(b[0] - ollf) / abs(ollf), reml_vs_oracle=abs(b[2] - orl) / abs(orl),
                mu_vs_small=np.abs(b[4] - s[4]).max(), mu_vs_oracle=np.abs(b[4] - omu).max(), mse_vs_oracle=(np.abs(b[5] - omse) / np.maximum(np.abs(omse), 1e-12 * float(st.sigma2[0]))).max())
    ok = errs["llf_vs_small"] < 1e-11 and errs["grad_vs_small"] < 1e-8 and errs["reml_vs_small"] < 1e-11 and errs["llf_vs_oracle"] < 1e-9 and errs["reml_vs_oracle"] < 1e-9 and errs["mu_vs_oracle"] < 1e-8 and errs["mse_vs_oracle"] < 1e-6
    print("N=%d d=%d trend=%d: " % (N, d, trend) + "  ".join("%s %.1e" % kv for kv in errs.items()), "" if ok else "  <-- FAIL", flush=True)
    bad += not ok
print("big-path trend sizes: %d failures" % bad)
sys.exit(1 if bad else 0)
