"""Stability soak: many set_train / fit-evaluation / commit / sweep cycles with changing sizes on one handle, and many
handles created and destroyed; device memory in use must return to its starting level (no leaks), results stay finite."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bogp import _lib  # noqa: E402


def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def main():
    rng = np.random.default_rng(0)
    eng = _lib.Engine(0)
    base = None
    t0 = time.perf_counter()
    for it in range(120):
        d = int(rng.integers(1, 24))
        N = int(rng.integers(d + 3, 700))
        n_t = 1 if it % 7 else 2
        X = rng.uniform(-5, 5, (N, d))
        y = rng.standard_normal((N, n_t))
        trend = 0 if n_t > 1 else int(rng.integers(0, 2))
        est = bool(rng.integers(0, 2)) and n_t == 1
        par = np.r_[np.full(d, 0.3 / d), 0.9]
        eng.set_train(X, y)
        beta = 0.0 if (est or trend == 0) else np.zeros(d + 1)
        try:
            eng.nll(int(rng.integers(0, 5)) if trend == 0 else 0, 1, par, 1e-3, est, beta, eval_grad=bool(it % 2), trend=trend)
        except _lib.NotPositiveDefinite:
            pass  # llf > 0 on a tiny random problem: the -inf convention
        eng.commit(3, 1, par, 1e-3, est, beta, trend=trend)
        M = int(rng.integers(1, 40000))
        eng.generate_candidates([-5.0] * d, [5.0] * d, M, seed=it, method=("uniform", "LHS", "sobol")[it % 3])
        best, idx = eng.sweep([(0, 0.0), (3, 2.0)], float(y[:, 0].min()), True)
        assert np.all(np.isfinite(best)) and np.all(idx >= 0) and np.all(idx < M)
        if trend == 0 and n_t == 1:
            mu, mse, dmu, dmse, v = eng.point_eval(X[0] + 0.1, [(0, 0.0)], float(y.min()), True)
            assert np.isfinite(mu) and mse >= 0 and np.all(np.isfinite(dmu))
            B = int(rng.integers(1, 70))  # r03: the batched flavour and the lock-step polish, sizes changing every cycle
            Xb = rng.uniform(-5, 5, (B, d))
            out = eng.point_eval_batch(Xb, [(0, 0.0), (2, 0.5)], float(y.min()), True)
            assert np.all(np.isfinite(out[0])) and np.all(out[1] >= 0) and out[5].shape == (B, 2, d)
            if d <= 64 and it % 3 == 0:
                Xp, fp, ne = eng.polish(Xb[: min(B, 16)], [-5.0] * d, [5.0] * d, (2, 0.5), float(y.min()), True, max_evals=12)
                assert np.all(np.isfinite(fp)) and np.all(Xp >= -5) and np.all(Xp <= 5)
        if trend == 0 and n_t == 1:  # r04: batched likelihoods of changing batch size, a lock-step MLE, a lazily uploaded sweep
            P = int(rng.integers(1, 24))
            pars = np.tile(par, (P, 1)) * 10.0 ** rng.uniform(-0.3, 0.3, size=(P, d + 1))
            if it % 5 == 0:
                pars[P // 2, 0] = np.nan  # an invalid slot now and then
            kb = int(rng.integers(0, 5))
            l, g, info = eng.nll_batch(kb, 1, pars, 1e-3, est, 0.0, eval_grad=bool(it % 2))
            assert np.all(np.isfinite(l[info == 0])) and (it % 5 != 0 or info[P // 2] == _lib.ERR_INVALID)
            if it % 4 == 0:
                lo_, hi_ = np.r_[np.full(d, -3.0), -5.0], np.r_[np.full(d, 1.0), 0.0]
                xo, fo, ne, st_, rounds = eng.mle_batch(kb, 1, rng.uniform(lo_, hi_, size=(int(rng.integers(1, 9)), d + 1)), lo_, hi_, 1e-3, est, 0.0,
                                                        eval_budget=int(rng.integers(20, 200)), prune_reserve=int(rng.integers(0, 2)) * 10)  # fmt: skip
                assert np.all(xo >= lo_) and np.all(xo <= hi_) and rounds <= ne.sum()
            eng.commit(3, 1, par, 1e-3, est, beta, trend=trend)
            Xh = rng.uniform(-5, 5, (int(rng.integers(1, 60000)), d))
            eng.upload_candidates(Xh, lazy=True)
            b2, i2 = eng.sweep([(0, 0.0)], float(y[:, 0].min()), True)
            assert np.isfinite(b2[0]) and 0 <= i2[0] < len(Xh)
        if it == 20:
            base = used_mb()
    mid = used_mb()
    for _ in range(40):
        e2 = _lib.Engine(0)
        e2.set_train(rng.uniform(-1, 1, (300, 4)), rng.standard_normal((300, 1)))
        e2.commit(0, 1, np.r_[np.full(4, 0.5), 0.9], 1e-6)
        e2.close()
    end = used_mb()
    print("soak: 120 cycles in %.1f s; device memory in use after cycle 20: %.0f MB, after 120: %.0f MB, after 40 create/destroy: %.0f MB" % (time.perf_counter() - t0, base, mid, end))
    assert mid - base < 2200 and abs(end - mid) < 64, "device memory grew"
    eng.close()
    print("soak ok; after closing the handle: %.0f MB" % used_mb())


if __name__ == "__main__":
    main()
