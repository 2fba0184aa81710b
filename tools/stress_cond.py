"""Ill-conditioning stress: how far do the explicit-inverse contraction (device) and LAPACK's triangular solve (oracle)
drift apart as cond(R) grows?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bogp import _lib
from oracle import gp_oracle as O
eng = _lib.Engine(0)
rng = np.random.default_rng(0)
N, d = 300, 2
X = rng.uniform(-5, 5, size=(N, d)); y = np.sin(X[:, :1]) + 0.1 * X[:, 1:] ** 2; y = (y - y.mean()) / y.std()
Xs = rng.uniform(-5, 5, size=(2000, d))
for nug in (1e-2, 1e-4, 1e-6, 1e-8, 1e-10):
    for th in (1.0, 0.1, 0.02):
        par = np.r_[np.full(d, th), 0.9]
        try:
            st = O.make_state(par, X, y, 0, 1, nug)
        except Exception as e:
            print("nug %.0e theta %.2f: oracle rejects (%s)" % (nug, th, str(e)[:40])); continue
        cond = np.linalg.cond(st.C) ** 2
        eng.set_train(X, y)
        try:
            eng.commit(0, 1, par, nug)
        except Exception as e:
            print("nug %.0e theta %.2f cond %.1e: device rejects (%s)" % (nug, th, cond, str(e)[:50])); continue
        eng.upload_candidates(Xs); mu, mse = eng.predict(); rmu, rmse = O.predict(st, Xs)
        print("nug %.0e theta %.2f cond(R) %.1e: max|dmu| %.1e (|mu| %.1e)  max|dmse| %.1e  (sigma2 0.9; min mse %.1e)" % (nug, th, cond, np.abs(mu - rmu.ravel()).max(), np.abs(rmu).max(), np.abs(mse - rmse.ravel()).max(), rmse.min()))
