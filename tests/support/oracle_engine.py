"""TEST-ONLY stand-in for `bogp._lib.Engine` backed by the CPU oracle.

Purpose: exercise the HOST logic of the drop-in layer (protocol plumbing with the real `bayes_optim` drivers, the
MLE restart loop, pickling) in the build container, which has no GPU.  It is never importable from the product
package and nothing in the product path can reach it; GPU parity is established separately by `-m gpu` tests."""
import numpy as np

from bogp import _lib
from oracle import gp_oracle as O


class OracleEngine:
    def __init__(self, device=0):
        self.N = self.d = self.M = 0
        self.st = None

    def close(self):
        pass

    def set_train(self, X, y):
        self.X, self.y = np.asarray(X, float), np.asarray(y, float).reshape(len(X), -1)
        self.N, self.d = self.X.shape
        self.n_t, self.target = self.y.shape[1], 0
        self.st = None

    def select_target(self, t):
        assert 0 <= t < self.n_t
        self.target = int(t)

    def nll(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=0):
        out = O.log_likelihood_concentrated(par, self.X, self.y, kernel, mode, noise_var, trend, estimate_trend, beta, eval_grad=eval_grad)
        llf = out[0] if eval_grad else out
        if not np.isfinite(llf):
            raise _lib.NotPositiveDefinite(_lib.ERR_NOT_POSDEF, "oracle: -inf")
        return (out[0], np.asarray(out[1], float).ravel()) if eval_grad else out

    def nll_batch(self, kernel, mode, pars, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=0):
        pars = np.atleast_2d(np.asarray(pars, float))
        llf, grad, info = np.empty(len(pars)), (np.zeros(pars.shape) if eval_grad else None), np.zeros(len(pars), dtype=np.int32)
        for s, p in enumerate(pars):
            if not (np.all(np.isfinite(p)) and np.all(p > 0)):
                llf[s], info[s] = -np.inf, _lib.ERR_INVALID
                continue
            try:
                out = self.nll(kernel, mode, p, noise_var, estimate_trend, beta, eval_grad=eval_grad, trend=trend)
                llf[s] = out[0] if eval_grad else out
                if eval_grad:
                    grad[s] = out[1]
            except _lib.NotPositiveDefinite as e:
                llf[s], info[s] = -np.inf, e.code
        return llf, grad, info

    def mle_batch(self, kernel, mode, x0, lo, hi, noise_var=0.0, estimate_trend=False, beta=0.0, trend=0, restricted=False, eval_budget=0,
                  m=10, factr=1e7, pgtol=1e-5, chain_rule=False, prune_reserve=0):
        """bogp_mle_batch's contract on the CPU: the library's OWN optimiser (bogp_lbfgsb_minimize, no device needed) on the oracle's
        likelihood, one start after the other (the lock step only changes how evaluations are grouped, not what a run does as long as the
        shared budget does not bind)."""
        x0 = np.atleast_2d(np.asarray(x0, float))
        fun = self.nll_restricted if restricted else self.nll

        def obj(x):
            par = 10.0 ** np.asarray(x)
            try:
                llf, g = fun(kernel, mode, par, noise_var, estimate_trend, beta, eval_grad=True, trend=trend)
            except _lib.NotPositiveDefinite:
                return np.inf, np.zeros(len(par))
            if not np.isfinite(llf):
                return np.inf, -np.asarray(g, float).ravel()
            g = -np.asarray(g, float).ravel()
            return -llf, (g * np.log(10.0) * par if chain_rule else g)

        R = len(x0)
        xopt, fopt = np.empty_like(x0), np.empty(R)
        nev, status = np.zeros(R, dtype=np.int32), np.zeros(R, dtype=np.int32)
        left = int(eval_budget) if eval_budget > 0 else 15000 * R
        for r in range(R):
            xopt[r], fopt[r], info = _lib.lbfgsb_minimize(obj, x0[r], np.c_[lo, hi], m=m, factr=factr, pgtol=pgtol, maxfun=max(1, left // (R - r)))
            nev[r], status[r] = info["funcalls"], info["status"]
            left -= info["funcalls"]
        return xopt, fopt, nev, status, int(nev.max())

    def nll_restricted(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=0):
        out = O.log_likelihood_restricted(par, self.X, self.y, kernel, mode, noise_var, trend, estimate_trend, beta, eval_grad=eval_grad)
        return (out[0], np.asarray(out[1], float).ravel()) if eval_grad else out

    def commit(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, trend=0):
        try:
            self.st = O.make_state(par, self.X, self.y, kernel, mode, noise_var, trend=trend, estimate_trend=estimate_trend, beta=beta)
        except np.linalg.LinAlgError as e:
            raise _lib.NotPositiveDefinite(_lib.ERR_NOT_POSDEF, str(e))
        self.target = 0
        return self.st.llf

    def get_state(self, with_C=True):
        st = self.st
        z = np.zeros(self.N)
        if self.n_t > 1:  # the shapes bogp._lib.Engine.get_state stacks for several targets
            return dict(C=st.C, gamma=st.gamma, rho=st.rho, Yt=st.Yt, Ft=z, Q=z, G=0.0, beta=float(st.beta[0, 0]),
                        sigma2=st.sigma2, noise_var=np.broadcast_to(np.asarray(st.noise_var, float).ravel(), (self.n_t,)).copy())  # fmt: skip
        if st.trend != 0:  # p > 1: matrices, as bogp_get_trend_state returns them
            return dict(C=st.C, gamma=st.gamma.ravel(), rho=st.rho.ravel(), Yt=st.Yt.ravel(), Ft=st.Ft, Q=st.Q, G=st.G,
                        beta=st.beta.ravel(), sigma2=float(st.sigma2[0]), noise_var=st.noise_var)  # fmt: skip
        return dict(C=st.C, gamma=st.gamma.ravel(), rho=st.rho.ravel(), Yt=st.Yt.ravel(),
                    Ft=z if st.Ft is None else st.Ft.ravel(), Q=z if st.Q is None else st.Q.ravel(),
                    G=0.0 if st.G is None else float(st.G[0, 0]), beta=float(st.beta[0, 0]), sigma2=float(st.sigma2[0]),
                    noise_var=st.noise_var)  # fmt: skip

    def upload_candidates(self, Xs, lazy=False):
        self.Xs = np.asarray(Xs, float)
        self.M = len(self.Xs)

    def predict(self, eval_MSE=True):
        mu, mse = O.predict_chunked(self.st, self.Xs, 1024)
        return mu[:, self.target].copy(), (mse[:, self.target].copy() if eval_MSE else None)

    def _vals(self, acq, plugin, minimize):
        mu, mse = self.predict()
        return [O.acquisition(a, p, mu, mse, plugin, self.st.sigma2[0], minimize) for a, p in acq]

    def sweep(self, acq, plugin, minimize=True, return_values=False, local_result=True):
        vals = self._vals(acq, plugin, minimize)
        idx = np.array([int(np.argmax(v)) for v in vals], dtype=np.int64)
        best = np.array([v[i] for v, i in zip(vals, idx)])
        self._last = (best, idx)
        if not local_result:  # bogp_sweep with NULL outputs: the winners stay "on the device" for exchange_argmax
            return None
        return (best, idx, np.array(vals)) if return_values else (best, idx)

    def read_candidates(self, rows):
        return self.Xs[np.asarray(rows, dtype=np.int64)].copy()

    def last_timing(self):
        return dict(corr_ms=0.0, contract_ms=0.0, acquisition_ms=0.0, n_chunks=0)

    # -- the cross-rank exchange of bogp_comm.hip, on torch.distributed (gloo) instead of RCCL: same records, same reduce ------------
    comm_rank, comm_world = 0, 0

    @staticmethod
    def comm_unique_id():
        return b"oracle-engine-stand-in".ljust(_lib.COMM_ID_BYTES, b"\0")

    def comm_init(self, uid, rank, world):
        assert len(uid) == _lib.COMM_ID_BYTES
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        self.comm_rank, self.comm_world = 0, 0

    def exchange_argmax(self, q, index_offset, with_x=True):
        from bogp import distributed

        assert self.comm_world > 0, "exchange without a communicator"
        best, idx = self._last
        assert len(best) == q
        return distributed.exchange_argmax(best, idx + int(index_offset), self.Xs[idx] if with_x else None)

    def sweep_topk(self, acq, plugin, minimize=True, k=1):
        vals = self._vals(acq, plugin, minimize)
        best = np.full((len(acq), k), -np.inf)
        idx = np.full((len(acq), k), -1, dtype=np.int64)
        for c, v in enumerate(vals):
            order = np.lexsort((np.arange(len(v)), -v))[:k]
            best[c, : len(order)], idx[c, : len(order)] = v[order], order
        return best, idx

    def gradient(self, x):
        a, b = O.gradient(self.st, x)
        return a.ravel(), b.ravel()
