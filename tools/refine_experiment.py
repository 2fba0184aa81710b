"""does one Newton-Schulz step on the explicit inverse (BOGP_REFINE_V=1) bring the ill-conditioned rows of
tools/stress_cond.py back under 1e-6?  Both the device and the LAPACK-backed oracle carry cond(R) eps of error there, so the
judge of this experiment is an 80-bit (np.longdouble) Cholesky + substitution of the same posterior mean.
  python tools/refine_experiment.py        (run twice: without and with BOGP_REFINE_V=1)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bogp import _lib  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def ld_posterior_mean(R, y, r):
    """mu = r R^-1 y in extended precision (simple kriging, beta = 0): column Cholesky + two substitutions."""
    L = np.array(R, dtype=np.longdouble)
    n = len(L)
    for j in range(n):
        L[j, j] = np.sqrt(L[j, j] - np.dot(L[j, :j], L[j, :j]))
        L[j + 1 :, j] = (L[j + 1 :, j] - L[j + 1 :, :j] @ L[j, :j]) / L[j, j]
    L = np.tril(L)
    z = np.array(y, dtype=np.longdouble).ravel().copy()
    for i in range(n):
        z[i] = (z[i] - np.dot(L[i, :i], z[:i])) / L[i, i]
    for i in range(n - 1, -1, -1):
        z[i] = (z[i] - np.dot(L[i + 1 :, i], z[i + 1 :])) / L[i, i]
    return np.asarray(np.array(r, dtype=np.longdouble) @ z, dtype=np.float64)


eng = _lib.Engine(0)
rng = np.random.default_rng(0)
N, d = 300, 2
X = rng.uniform(-5, 5, size=(N, d))
y = np.sin(X[:, :1]) + 0.1 * X[:, 1:] ** 2
y = (y - y.mean()) / y.std()
Xs = rng.uniform(-5, 5, size=(400, d))
print("BOGP_REFINE_V=%s BOGP_REFINE_GAMMA=%s" % (os.environ.get("BOGP_REFINE_V", "0"), os.environ.get("BOGP_REFINE_GAMMA", "1 (default)")))
for nug in (1e-6, 1e-8, 1e-10, 1e-11):
    par = np.r_[np.full(d, 0.02), 0.9]
    try:
        st = O.make_state(par, X, y, 0, 1, nug)
    except Exception as e:  # noqa: BLE001
        print("nug %.0e: oracle rejects (%s)" % (nug, str(e)[:40]))
        continue
    s2, nv = 0.9, nug
    R0 = O.correlation_matrix(0, par[:d], X)
    Rn = (s2 * R0 + nv * np.eye(N)) / (s2 + nv)  # the NOISY-mode matrix the reference factorises (gpr.py:963-969)
    r0 = O.corr(0, par[:d], O.l1_cross_distances(Xs, X)).reshape(len(Xs), N)  # un-normalised r against the normalised R: the reference's quirk
    truth = ld_posterior_mean(Rn, st.Yt * 0 + y, r0)
    eng.set_train(X, y)
    eng.commit(0, 1, par, nug)
    eng.upload_candidates(Xs)
    mu, _ = eng.predict()
    omu, _ = O.predict(st, Xs)
    cond = np.linalg.cond(st.C) ** 2
    sc = np.abs(truth).max()
    print("nug %.0e cond(R) %.1e: device vs 80-bit %.1e | oracle (LAPACK) vs 80-bit %.1e | device vs oracle %.1e   (relative to max|mu| = %.2f)"
          % (nug, cond, np.abs(mu - truth).max() / sc, np.abs(omu.ravel() - truth).max() / sc, np.abs(mu - omu.ravel()).max() / sc, sc))
