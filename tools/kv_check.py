"""r05: the device's K_nu, 1 / Gamma and general-nu Matern profile against mpmath (40 digits) and scipy on random (nu, x), and the
closed-form profiles against the reference's numpy expressions.  Prints eps statistics (-> profiles/r05_kv_accuracy.txt).
Usage (GPU box): python tools/kv_check.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpmath as mp
import numpy as np
from scipy.special import gamma, kv

from bogp import _lib

mp.mp.dps = 40
EPS = 2.0**-52
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(55)
eng = _lib.Engine(0)


def stats(tag, e):
    e = np.asarray(e)
    print("%-58s max %8.2f eps   99.9 %% %8.2f   median %.2f" % (tag, e.max(), np.quantile(e, 0.999), np.median(e)))


for lo, hi in ((1e-8, 1e-3), (1e-3, 0.5), (0.5, 1.0), (1.0, 2.0), (2.0, 10.0), (10.0, 700.0)):
    nus = np.r_[rng.uniform(1e-3, 10.0, 12), rng.choice([0.5, 1.0, 1.5, 2.5, 0.25, 3.7, 0.8, 10.0, 0.49999, 1.00001], 4)]
    ed, es, pd, ps = [], [], [], []
    for nu in nus:
        x = 10.0 ** rng.uniform(np.log10(lo), np.log10(hi), n // (6 * 16))
        dev = eng.selftest_profile(_lib.SELFTEST_BESSEL_K, x, pexp=nu)
        # the profile's argument is s2 with tmp = sqrt(2 nu) sqrt(s2): hand it s2 = x^2 / (2 nu) and compare at the tmp the device formed
        s2 = x * x / (2.0 * nu)
        tmp = np.sqrt(2.0 * nu) * np.sqrt(s2)
        prof = eng.selftest_profile(_lib.SELFTEST_PROFILE, s2, kernel=_lib.KERNEL_MATERN_NU, pexp=nu)
        sc = kv(nu, x)
        scp = 2.0 ** (1.0 - nu) / gamma(nu)
        scp = scp * tmp**nu
        scp = scp * kv(nu, tmp)
        for i in range(len(x)):
            t = mp.besselk(mp.mpf(nu), mp.mpf(float(x[i])))
            ed.append(float(abs((mp.mpf(float(dev[i])) - t) / t)) / EPS)
            es.append(float(abs((mp.mpf(float(sc[i])) - t) / t)) / EPS)
            tt = mp.mpf(float(tmp[i]))
            tr = mp.mpf(2) ** (1 - mp.mpf(nu)) / mp.gamma(mp.mpf(nu)) * tt ** mp.mpf(nu) * mp.besselk(mp.mpf(nu), tt)
            if tr > 1e-290:
                pd.append(float(abs((mp.mpf(float(prof[i])) - tr) / tr)) / EPS)
                ps.append(float(abs((mp.mpf(float(scp[i])) - tr) / tr)) / EPS)
    print("x in [%g, %g]" % (lo, hi))
    stats("  device K_nu vs truth", ed)
    stats("  scipy.special.kv vs truth", es)
    stats("  device Matern-nu profile vs truth", pd)
    stats("  reference's expression (scipy kv, gamma) vs truth", ps)
z = rng.uniform(1e-6, 12.0, 2000)
rg = eng.selftest_profile(_lib.SELFTEST_RGAMMA, z)
stats("device 1 / Gamma(nu), nu in (0, 12] vs truth", [float(abs(mp.mpf(float(a)) * mp.gamma(mp.mpf(float(b))) - 1)) / EPS for a, b in zip(rg, z)])
stats("scipy.special.gamma vs truth", [float(abs(mp.mpf(float(gamma(b))) / mp.gamma(mp.mpf(float(b))) - 1)) / EPS for b in z])
for kid, name in ((_lib.KERNEL_SE, "SE"), (_lib.KERNEL_MATERN12, "Matern-1/2"), (_lib.KERNEL_MATERN32, "Matern-3/2"), (_lib.KERNEL_MATERN52, "Matern-5/2")):
    s2 = np.r_[rng.uniform(0, 40.0, 500_000), 10.0 ** rng.uniform(-20, 2.5, 500_000), 0.0]
    got = eng.selftest_profile(_lib.SELFTEST_PROFILE, s2, kernel=kid)
    if kid == _lib.KERNEL_SE:
        want = np.exp(-s2)
    else:
        D = np.sqrt(s2)
        if kid == _lib.KERNEL_MATERN12:
            want = np.exp(-D)
        elif kid == _lib.KERNEL_MATERN32:
            K = D * np.sqrt(3.0)
            want = (1.0 + K) * np.exp(-K)
        else:
            K = D * np.sqrt(5.0)
            want = (1.0 + K + K**2 / 3.0) * np.exp(-K)
    ok = want > 1e-300
    print("%-11s profile vs the reference's numpy expression: max %.3f ulp, %.3f %% of 1e6 values differ; r(0) = %r" % (
        name, np.max(np.abs(got - want)[ok] / np.spacing(want)[ok]), 100.0 * np.mean(got[ok] != want[ok]), got[-1]))
eng.close()
