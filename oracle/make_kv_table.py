"""K_nu(x) and the general-nu Matern profile on 1e5 (nu, x) pairs: the TRUE values (mpmath, 40 digits) beside what the reference computes
(`scipy.special.kv`, `scipy.special.gamma`, kernel.py:201-207) -> tests/golden/G36_kv_table.npz.

Run (build container):  python oracle/make_kv_table.py        (~10 min, 4 processes)

Why a truth table and not only scipy's values: scipy's kv (AMOS zbesk) is itself up to ~500 eps from the true value on this domain
(the table's `kv_scipy_err_eps` column), so "the device agrees with scipy to a few eps" cannot hold for an accurate K_nu -- what can
be asserted per pair is  |device - truth| <= a few eps  and  |device - scipy| <= |scipy - truth| + a few eps.
Domain: nu in (0, 10] (a fifth of the pairs on / a hair beside integers and half-integers), x in [1e-8, 700] log-uniform plus
clusters at the method boundaries x = 1 and x = 2.  `profile_*`: r = 2^(1-nu) / Gamma(nu) x^nu K_nu(x)  (kernel.py:204-207 with tmp = x)."""
import os
import sys
from multiprocessing import Pool

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "G36_kv_table.npz")
N = 100_000


def pairs():
    rng = np.random.default_rng(36)
    nu = rng.uniform(0.0, 10.0, N)
    special = np.r_[np.arange(1, 21) / 2.0, 0.25, 0.75, 1e-3, 1e-6, 3.7, 0.8]
    k = N // 5
    nu[:k] = rng.choice(special, k) * (1.0 + rng.choice([0.0, 0.0, 1e-5, -1e-5, 1e-12, -1e-12], k))
    nu = np.clip(nu, 1e-6, 10.0)
    x = 10.0 ** rng.uniform(-8.0, np.log10(700.0), N)
    x[k : k + 2000] = 1.0 + rng.uniform(-1e-3, 1e-3, 2000)
    x[k + 2000 : k + 4000] = 2.0 + rng.uniform(-1e-3, 1e-3, 2000)
    x[k + 4000 : k + 4100] = [1.0, 2.0, 1.0 - 2.0**-53, 1.0 + 2.0**-52] * 25
    return nu, x


def truth(args):
    import mpmath as mp

    mp.mp.dps = 40
    nu, x = args
    kv, pr = np.empty(len(nu)), np.empty(len(nu))
    kv_lo, pr_lo = np.empty(len(nu)), np.empty(len(nu))
    for i, (a, b) in enumerate(zip(nu, x)):
        a, b = mp.mpf(float(a)), mp.mpf(float(b))
        t = mp.besselk(a, b)
        r = mp.mpf(2) ** (1 - a) / mp.gamma(a) * b**a * t
        kv[i], pr[i] = float(t), float(r)
        kv_lo[i], pr_lo[i] = float(t - mp.mpf(kv[i])), float(r - mp.mpf(pr[i]))  # truth = hi + lo to ~32 digits
    return kv, kv_lo, pr, pr_lo


if __name__ == "__main__":
    import scipy
    from scipy.special import gamma, kv

    nu, x = pairs()
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    chunks = [(nu[i::nproc], x[i::nproc]) for i in range(nproc)]
    with Pool(nproc) as pool:
        res = pool.map(truth, chunks)
    K, Kl, P, Pl = (np.empty(N) for _ in range(4))
    for i, (a, b, c, d) in enumerate(res):
        K[i::nproc], Kl[i::nproc], P[i::nproc], Pl[i::nproc] = a, b, c, d
    ks = kv(nu, x)
    ps = 2.0 ** (1.0 - nu) / gamma(nu)
    ps = ps * x**nu
    ps = ps * ks
    eps = 2.0**-52
    with np.errstate(all="ignore"):
        kerr = np.abs((ks - K) - Kl) / np.abs(K) / eps
        perr = np.abs((ps - P) - Pl) / np.abs(P) / eps
    ok = np.isfinite(K) & (np.abs(K) > 1e-290) & (np.abs(P) > 1e-290)
    print("scipy.special.kv against the truth:   max %.1f eps, 99.9 %% %.1f, median %.2f" % (kerr[ok].max(), np.quantile(kerr[ok], 0.999), np.median(kerr[ok])))
    print("the reference's profile expression:  max %.1f eps, 99.9 %% %.1f, median %.2f" % (perr[ok].max(), np.quantile(perr[ok], 0.999), np.median(perr[ok])))
    # (nu, x) are regenerated from the seed by the reader (pairs()); truth = double (1 + rlo), the relative residual as float32
    np.savez_compressed(OUT, n=N, seed=36, nu_x_checksum=float(np.sum(nu) + np.sum(x)), kv_true=K, kv_true_rlo=(Kl / K).astype(np.float32), profile_true=P,
                        profile_true_rlo=np.where(P != 0, Pl / np.where(P != 0, P, 1.0), 0.0).astype(np.float32), kv_scipy=ks, scipy=scipy.__version__, numpy=np.__version__)
    print("wrote", OUT)
