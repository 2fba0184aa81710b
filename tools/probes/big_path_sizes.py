"""The 128-tile fit path (every N > 3072 since r06) against the 64-block kernels (BOGP_NO_BIG_FIT=1) and the NumPy oracle at awkward sizes:
leading dimensions with 1 ... 127 rows of identity padding, tile counts that are odd / prime, both sides of the wide-panel threshold."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bogp import _lib
from oracle import gp_oracle as O

sizes = [int(a) for a in sys.argv[1:]] or [3073, 3100, 3199, 3200, 3201, 3585, 4000, 4097, 5000, 5555, 6016, 6017, 6079, 6081, 6500, 7553]
bad = 0
for N in sizes:
    d = 5
    rng = np.random.default_rng(N)
    X = rng.uniform(-5, 5, size=(N, d)); y = np.sum(X**2, axis=1)
    y = ((y - y.mean()) / y.std() + 0.2 * rng.standard_normal(N)).reshape(-1, 1)
    par = np.r_[np.full(d, 0.25) * rng.uniform(0.7, 1.3, size=d), 0.8]
    out = {}
    for tag, flag in (("big", None), ("small", "1")):
        os.environ.pop("BOGP_NO_BIG_FIT", None)
        if flag: os.environ["BOGP_NO_BIG_FIT"] = flag
        eng = _lib.Engine(0); eng.set_train(X, y)
        kid = _lib.KERNEL_MATERN52 if N % 2 else _lib.KERNEL_SE
        llf, grad = eng.nll(kid, _lib.MODE_NOISY, par, 1e-6, True, 0.0, eval_grad=True)
        eng.commit(kid, _lib.MODE_NOISY, par, 1e-6, True, 0.0)
        Xs = rng.uniform(-5, 5, size=(257, d)) if tag == "big" else Xs
        eng.upload_candidates(Xs); mu, mse = eng.predict()
        out[tag] = (llf, grad, mu, mse); eng.close()
    os.environ.pop("BOGP_NO_BIG_FIT", None)
    b, s = out["big"], out["small"]
    e_llf = abs(b[0] - s[0]) / abs(s[0]); e_g = np.abs(b[1] - s[1]).max() / np.abs(s[1]).max()
    e_mu = np.abs(b[2] - s[2]).max(); e_mse = (np.abs(b[3] - s[3]) / np.maximum(np.abs(s[3]), 1e-12)).max()
    msg = "N=%d (ld %d): big vs 64-block  llf %.1e  grad %.1e  mu %.1e  mse %.1e" % (N, -(-N // 128) * 128, e_llf, e_g, e_mu, e_mse)
    ok = e_llf < 1e-11 and e_g < 1e-8 and e_mu < 1e-9 and e_mse < 1e-6
    if N <= 4100:  # the oracle's dense factorisation is a few seconds here
        okid = O.KERNEL_MATERN52 if N % 2 else O.KERNEL_SE
        ollf = float(O.log_likelihood_concentrated(par, X, y, okid, O.MODE_NOISY, 1e-6, estimate_trend=True, beta=0.0))
        e_o = abs(b[0] - ollf) / abs(ollf); msg += "  | vs oracle llf %.1e" % e_o; ok = ok and e_o < 1e-9
    print(msg, "" if ok else "  <-- FAIL", flush=True)
    bad += not ok
print("big-path sizes: %d sizes, %d failures" % (len(sizes), bad))
sys.exit(1 if bad else 0)
