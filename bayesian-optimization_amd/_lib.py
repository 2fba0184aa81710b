"""ctypes binding of libbogp.so (include/bogp.h).  There is no fallback: if the library is missing or no
gfx950 device is usable, every entry point raises -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Optional, Sequence, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbogp.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_POSDEF, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_LLF_POSITIVE = -1, -2, -3, -4, -5, -6
KERNEL_SE, KERNEL_MATERN12, KERNEL_MATERN32, KERNEL_MATERN52, KERNEL_ABSEXP, KERNEL_CUBIC, KERNEL_GENEXP, KERNEL_MATERN_NU = 0, 1, 2, 3, 4, 5, 6, 7
MODE_NOISELESS, MODE_NOISY, MODE_NOISE_ESTIM = 0, 1, 2
ACQ_EI, ACQ_EPSILON_PI, ACQ_UCB, ACQ_MGFI = 0, 1, 2, 3
TREND_CONSTANT, TREND_LINEAR, TREND_QUADRATIC = 0, 1, 2
SELFTEST_PROFILE, SELFTEST_BESSEL_K, SELFTEST_RGAMMA, SELFTEST_BESSEL_K_PAIRS, SELFTEST_MATERN_NU_PAIRS = range(5)
ABI_VERSION = 9  # BOGP_ABI_VERSION of include/bogp.h this binding table was written for
MAX_Q = 64
MAX_TARGETS = 8
COMM_ID_BYTES = 128
SCALES = {"linear": 0, None: 0, "log": 1, "log10": 2, "logit": 3, "bilog": 4}  # Real.scale (variable.py:43-55)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)

# name -> (restype, argtypes); must list every symbol include/bogp.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "bogp_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "bogp_destroy": (None, [C.c_void_p]),
    "bogp_last_error": (C.c_char_p, [C.c_void_p]),
    "bogp_abi_version": (C.c_int, []),
    "bogp_set_train": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_int]),
    "bogp_select_target": (C.c_int, [C.c_void_p, C.c_int]),
    "bogp_nll": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp, _dp]),
    "bogp_nll_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp, _dp, _ip]),
    "bogp_mle_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, C.c_double,
                                 C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _ip]),
    "bogp_lbfgsb_minimize": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       _dp, _ip, _ip, _ip]),
    "bogp_commit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp]),
    "bogp_nll_restricted": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, _dp, _dp]),
    "bogp_get_state": (C.c_int, [C.c_void_p] + [_dp] * 10),
    "bogp_trend_size": (C.c_int, [C.c_int, C.c_int]),
    "bogp_set_trend_beta": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "bogp_get_trend_state": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "bogp_candidates_upload": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "bogp_candidates_upload_lazy": (C.c_int, [C.c_void_p, _dp, C.c_int64]),
    "bogp_candidates_bind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "bogp_candidates_generate": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_uint64, C.c_int64]),
    "bogp_candidates_generate_lhs": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_uint64, C.c_int64, C.c_int64]),
    "bogp_candidates_generate_lhs_maximin": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_uint64, C.c_int, _dp, _ip]),
    "bogp_candidates_min_pdist2": (C.c_int, [C.c_void_p, _dp]),
    "bogp_candidates_generate_sobol": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int64, C.c_int64, C.POINTER(C.c_uint64), C.c_int]),
    "bogp_candidates_read": (C.c_int, [C.c_void_p, _lp, C.c_int, _dp]),
    "bogp_candidates_set_transform": (C.c_int, [C.c_void_p, _ip, _ip, _dp, _dp]),
    "bogp_predict": (C.c_int, [C.c_void_p, _dp, _dp]),
    "bogp_sweep": (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, C.c_double, C.c_int, _dp, _lp, _dp]),
    "bogp_sweep_topk": (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, C.c_double, C.c_int, C.c_int, _dp, _lp]),
    "bogp_gradient": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "bogp_gradient_batch": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    "bogp_hessian": (C.c_int, [C.c_void_p, _dp, _dp]),
    "bogp_prior_corr": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    "bogp_point_eval": (C.c_int, [C.c_void_p, _dp, C.c_int, _ip, _dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "bogp_point_eval_batch": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _ip, _dp, C.c_double, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]),
    "bogp_polish": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
                              C.c_double, _dp, _dp, _ip]),
    "bogp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "bogp_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "bogp_comm_attach": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bogp_comm_info": (C.c_int, [C.c_void_p, _ip, _ip]),
    "bogp_comm_destroy": (C.c_int, [C.c_void_p]),
    "bogp_exchange_argmax": (C.c_int, [C.c_void_p, C.c_int64, _dp, _lp, _dp]),
    "bogp_exchange_topk": (C.c_int, [C.c_void_p, C.c_int64, _dp, _lp, _dp]),
    "bogp_reduce_pairs": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _lp, _dp]),
    "bogp_merge_topk": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _lp, _dp]),
    "bogp_last_timing": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _ip]),
    "bogp_flops_per_candidate": (C.c_double, [C.c_void_p]),
    "bogp_nll_path": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "bogp_chol_wide_panels": (C.c_int, [C.c_int, _ip, C.c_int]),
    "bogp_selftest_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, _dp, C.c_int64, _dp]),
    "bogp_selftest_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int, C.c_int, C.c_int]),
}

_lib = None


class BogpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libbogp error %d: %s" % (code, msg))
        self.code = code


class NotPositiveDefinite(BogpError, np.linalg.LinAlgError):
    """Cholesky info > 0 (or llf > 0, which the reference also rejects): the host maps it to llf = -inf."""


def load():
    """dlopen libbogp.so and bind every symbol.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libbogp.so not found at %s -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH
        )
    # Load order matters: the PyTorch wheel bundles its own libamdhip64 (the only ROCm library libbogp.so needs; it links
    # no BLAS) with the SAME soname as /opt/rocm's.  The dynamic loader keeps whichever copy comes first, and torch
    # crashes when it is handed the system copy; the other way round (libbogp on torch's copy) works.  So torch goes
    # first whenever it is installed (it provides torch.distributed for the multi-GPU exchange anyway).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table drift apart
        fn.restype = res
        fn.argtypes = args
    if lib.bogp_abi_version() != ABI_VERSION:
        raise ImportError("libbogp.so is at ABI %d, this binding at %d: rebuild it (make -C %s)"
                          % (lib.bogp_abi_version(), ABI_VERSION, os.path.join(os.path.dirname(LIB_PATH), "csrc")))
    _lib = lib
    return lib


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


OBJECTIVE_FN = C.CFUNCTYPE(None, _dp, C.c_int, _dp, _dp, C.c_void_p)


def lbfgsb_minimize(fun, x0, bounds, m=10, factr=1e7, pgtol=1e-5, maxfun=15000, maxiter=15000):
    """libbogp's L-BFGS-B (csrc/bogp_lbfgsb.h, the optimiser of the lock-step MLE) on a Python objective `fun(x) -> (f, g)`:
    the counterpart of scipy.optimize.fmin_l_bfgs_b(fun, x0, bounds=bounds) for tests.  No device is touched.
    Returns (x, f, {"funcalls", "nit", "status"})."""
    lib = load()
    x = np.array(x0, dtype=np.float64).ravel()
    n = len(x)
    b = np.asarray(bounds, dtype=np.float64).reshape(n, 2)
    lo, hi = np.ascontiguousarray(b[:, 0]), np.ascontiguousarray(b[:, 1])

    def cb(xp, nn, fp, gp, _user):
        xv = np.ctypeslib.as_array(xp, shape=(nn,)).copy()
        f, g = fun(xv)
        fp[0] = float(f)
        gv = np.asarray(g, dtype=np.float64).ravel()
        for i in range(nn):
            gp[i] = gv[i]

    cfn = OBJECTIVE_FN(cb)
    f, nfev, nit, status = C.c_double(), C.c_int(), C.c_int(), C.c_int()
    rc = lib.bogp_lbfgsb_minimize(n, int(m), _ptr(x), _ptr(lo), _ptr(hi), float(factr), float(pgtol), int(maxfun), int(maxiter),
                                  C.cast(cfn, C.c_void_p), None, C.byref(f), C.byref(nfev), C.byref(nit), C.byref(status))
    if rc != OK:
        raise BogpError(rc, "bogp_lbfgsb_minimize: invalid arguments")
    return x, f.value, {"funcalls": nfev.value, "nit": nit.value, "status": status.value}


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through libbogp (call on ONE rank, distribute the 128 bytes to the others)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load().bogp_comm_unique_id(buf)
    if rc != OK:
        raise BogpError(rc, "bogp_comm_unique_id failed (librccl missing?)")
    return buf.raw


def reduce_pairs_c(gathered: np.ndarray):
    """bogp_reduce_pairs on host records (R, q, 2 + d): (values (q,), global indices (q,), points (q, d) or None)."""
    g = _f64(gathered)
    R, q, rec = g.shape
    d = rec - 2
    val, idx = np.empty(q), np.empty(q, dtype=np.int64)
    x = np.empty((q, d)) if d else None
    rc = load().bogp_reduce_pairs(R, q, d, _ptr(g), _ptr(val), idx.ctypes.data_as(_lp), _ptr(x))
    if rc != OK:
        raise BogpError(rc, "bogp_reduce_pairs: invalid arguments")
    return val, idx, x


def merge_topk_c(gathered: np.ndarray):
    """bogp_merge_topk on host records (R, q, k, 2 + d): (values (q, k), global indices (q, k), points (q, k, d) or None)."""
    g = _f64(gathered)
    R, q, k, rec = g.shape
    d = rec - 2
    val, idx = np.empty((q, k)), np.empty((q, k), dtype=np.int64)
    x = np.empty((q, k, d)) if d else None
    rc = load().bogp_merge_topk(R, q, k, d, _ptr(g), _ptr(val), idx.ctypes.data_as(_lp), _ptr(x))
    if rc != OK:
        raise BogpError(rc, "bogp_merge_topk: invalid arguments")
    return val, idx, x


def trend_size_of(trend: int, d: int) -> int:
    """Columns of the trend basis (bogp_trend_size; needs no device)."""
    return 1 if trend == TREND_CONSTANT else (d + 1 if trend == TREND_LINEAR else (d + 1) * (d + 2) // 2)


@functools.lru_cache(maxsize=32)
def sobol_direction_numbers(d: int, bits: int = 30) -> np.ndarray:
    """(Cached per (d, bits): building the table costs 30 scipy engines, and generate_candidates(method="sobol") asks for it on
    every ask().  Needs scipy >= 1.9 for the `bits=` keyword of `qmc.Sobol`.  The returned array is shared: do not write to it.)
    The (d, bits) direction numbers of scipy's unscrambled Sobol' generator (Joe & Kuo tables): what the reference's
    `sobol_seq` call resolves to in this image.  Data for `bogp_candidates_generate_sobol`, which
    carries no table of its own.  Recovered through scipy's PUBLIC interface only: point n of the sequence is the XOR of
    the direction numbers over the set bits of the Gray code of n, and gray(2^b) = 2^b ^ 2^(b-1), so
    sv[b] = x_(2^b) ^ sv[b-1] with x_(2^b) read off `Sobol.fast_forward(2^b).random(1)` (exact multiples of 2^-bits)."""
    from scipy.stats import qmc

    sv = np.zeros((int(d), int(bits)), dtype=np.uint64)
    prev = np.zeros(int(d), dtype=np.uint64)
    for b in range(int(bits)):
        eng = qmc.Sobol(d=int(d), scramble=False, bits=int(bits))
        if b:
            eng.fast_forward(2**b)
        else:
            eng.fast_forward(1)
        ints = np.round(eng.random(1)[0] * 2.0 ** int(bits)).astype(np.uint64)
        sv[:, b] = ints ^ prev if b else ints
        prev = sv[:, b]
    return np.ascontiguousarray(sv)


class Engine:
    """One handle = one GPU.  Thin, stateful wrapper over the C ABI; raises BogpError on any non-zero code."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.bogp_create(int(device), C.byref(h))
        if rc != OK:
            raise BogpError(rc, (self._lib.bogp_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)
        self.N = self.d = 0
        self.trend, self.estimate_trend = TREND_CONSTANT, False
        self.M = 0
        self.comm_rank, self.comm_world = 0, 0  # world > 0 once a communicator is set (comm_init / comm_attach)
        self._keep = None  # keeps bound device memory owners alive

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bogp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc == OK:
            return
        msg = (self._lib.bogp_last_error(self._h) or b"").decode()
        if rc in (ERR_NOT_POSDEF, ERR_LLF_POSITIVE):
            raise NotPositiveDefinite(rc, msg)
        raise BogpError(rc, msg)

    # -- training set / likelihood / commit ---------------------------------------------------------
    def set_train(self, X, y):
        X = _f64(X)
        y = _f64(y).reshape(len(X), -1)
        self._check(self._lib.bogp_set_train(self._h, _ptr(X), _ptr(y), X.shape[0], X.shape[1], y.shape[1]))
        self.N, self.d = X.shape
        self.n_t = y.shape[1]

    def select_target(self, t: int):
        """Column of a multi-target y that get_state / predict / sweep / gradient work on (0 after set_train, commit)."""
        self._check(self._lib.bogp_select_target(self._h, int(t)))

    def _trend_beta(self, trend, estimate_trend, beta) -> float:
        """Fixed coefficients of a p > 1 basis travel through bogp_set_trend_beta; the scalar argument serves p = 1."""
        if trend == TREND_CONSTANT:
            return float(np.ravel(beta)[0]) if np.ndim(beta) else float(beta)
        if not estimate_trend:
            b = _f64(beta).ravel()
            self._check(self._lib.bogp_set_trend_beta(self._h, _ptr(b), len(b)))
        return 0.0

    def nll(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=TREND_CONSTANT):
        """log-likelihood (and d llf / d par) at `par`; raises NotPositiveDefinite where the reference returns -inf."""
        par = _f64(par).ravel()
        llf = C.c_double()
        grad = np.zeros(len(par)) if eval_grad else None
        b = self._trend_beta(trend, estimate_trend, beta)
        self._check(
            self._lib.bogp_nll(self._h, kernel, mode, _ptr(par), len(par), float(noise_var), int(trend),
                               int(bool(estimate_trend)), b, C.byref(llf), _ptr(grad))
        )  # fmt: skip
        return (llf.value, grad) if eval_grad else llf.value

    def nll_batch(self, kernel, mode, pars, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=TREND_CONSTANT):
        """P likelihood (+ gradient) evaluations in one device round trip: `pars` is (P, n_par).  Returns (llf (P,), grad (P, n_par)
        or None, info (P,) int32): llf[s] = -inf where the reference returns -inf (info[s] = ERR_NOT_POSDEF / ERR_LLF_POSITIVE /
        ERR_INVALID), with a zero gradient row.  Slot s holds the bits of the s-th sequential `nll` call."""
        pars = _f64(pars)
        if pars.ndim != 2:
            raise ValueError("pars must be (P, n_par)")
        P, n_par = pars.shape
        llf = np.empty(P)
        grad = np.zeros((P, n_par)) if eval_grad else None
        info = np.zeros(P, dtype=np.int32)
        b = self._trend_beta(trend, estimate_trend, beta)
        self._check(
            self._lib.bogp_nll_batch(self._h, kernel, mode, P, _ptr(pars), n_par, float(noise_var), int(trend),
                                     int(bool(estimate_trend)), b, _ptr(llf), _ptr(grad), info.ctypes.data_as(_ip))
        )  # fmt: skip
        llf[info != OK] = -np.inf
        return llf, grad, info

    def mle_batch(self, kernel, mode, x0, lo, hi, noise_var=0.0, estimate_trend=False, beta=0.0, trend=TREND_CONSTANT, restricted=False,
                  eval_budget=0, m=10, factr=1e7, pgtol=1e-5, chain_rule=False, prune_reserve=0):
        """The R restarts of the MLE in lock step (bogp_mle_batch): x0 (R, n_par) log10 starts, lo / hi (n_par,) log10 bounds.
        `prune_reserve` > 0 stops the worst run whenever fewer than that many evaluations per active run remain of a shared budget.
        `chain_rule=True` hands the optimiser the gradient w.r.t. log10(par) instead of the reference's un-scaled one (an extension).
        Returns (xopt (R, n_par) log10, fopt (R,) = -llf, n_evals (R,), status (R,), rounds)."""
        x0 = _f64(x0)
        if x0.ndim != 2:
            raise ValueError("x0 must be (R, n_par)")
        R, n_par = x0.shape
        lo, hi = _f64(lo).ravel(), _f64(hi).ravel()
        if len(lo) != n_par or len(hi) != n_par:
            raise ValueError("bounds must have n_par entries")
        xopt, fopt = np.empty((R, n_par)), np.empty(R)
        nev, status = np.zeros(R, dtype=np.int32), np.zeros(R, dtype=np.int32)
        rounds = C.c_int()
        b = self._trend_beta(trend, estimate_trend, beta)
        self._check(
            self._lib.bogp_mle_batch(self._h, kernel, mode, int(bool(restricted)), R, _ptr(x0), n_par, _ptr(lo), _ptr(hi), float(noise_var),
                                     int(trend), int(bool(estimate_trend)), b, int(eval_budget), int(m), float(factr), float(pgtol),
                                     1 if chain_rule else 0, int(prune_reserve), _ptr(xopt), _ptr(fopt), nev.ctypes.data_as(_ip), status.ctypes.data_as(_ip), C.byref(rounds))
        )  # fmt: skip
        return xopt, fopt, nev, status, rounds.value

    def nll_restricted(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, eval_grad=False, trend=TREND_CONSTANT):
        """Restricted (REML) log-likelihood, gpr.py:813-918.  exp(llf) > 1 gives -inf WITH the gradient of the finite
        value, as the reference returns it; a failed factorisation raises NotPositiveDefinite."""
        par = _f64(par).ravel()
        llf = C.c_double()
        grad = np.zeros(len(par)) if eval_grad else None
        b = self._trend_beta(trend, estimate_trend, beta)
        rc = self._lib.bogp_nll_restricted(self._h, kernel, mode, _ptr(par), len(par), float(noise_var), int(trend),
                                           int(bool(estimate_trend)), b, C.byref(llf), _ptr(grad))  # fmt: skip
        if rc == ERR_LLF_POSITIVE:
            return (-np.inf, grad) if eval_grad else -np.inf
        self._check(rc)
        return (llf.value, grad) if eval_grad else llf.value

    def commit(self, kernel, mode, par, noise_var=0.0, estimate_trend=False, beta=0.0, trend=TREND_CONSTANT) -> float:
        par = _f64(par).ravel()
        llf = C.c_double()
        b = self._trend_beta(trend, estimate_trend, beta)
        self._check(
            self._lib.bogp_commit(self._h, kernel, mode, _ptr(par), len(par), float(noise_var), int(trend),
                                  int(bool(estimate_trend)), b, C.byref(llf))
        )  # fmt: skip
        self.trend = int(trend)
        self.estimate_trend = bool(estimate_trend)
        return llf.value

    def trend_size(self, trend=None) -> int:
        return int(self._lib.bogp_trend_size(int(self.trend if trend is None else trend), int(self.d)))

    def get_state(self, with_C=True) -> dict:
        if getattr(self, "n_t", 1) > 1:  # one column per target: gamma, rho, Yt (N, n_t); sigma2, noise_var (n_t,)
            cols = []
            for t in range(self.n_t):
                self.select_target(t)
                cols.append(self._get_state_active(with_C and t == 0))
            self.select_target(0)
            out = dict(cols[0])
            for k in ("gamma", "rho", "Yt"):
                out[k] = np.column_stack([c[k] for c in cols])
            out["sigma2"] = np.array([c["sigma2"] for c in cols])
            out["noise_var"] = np.array([c["noise_var"] for c in cols])
            return out
        return self._get_state_active(with_C)

    def _get_state_active(self, with_C=True) -> dict:
        N = self.N
        Cm = np.empty((N, N)) if with_C else None
        v = {k: np.zeros(N) for k in ("gamma", "rho", "Yt", "Ft", "Q")}
        s = [C.c_double() for _ in range(4)]
        self._check(
            self._lib.bogp_get_state(self._h, _ptr(Cm), _ptr(v["gamma"]), _ptr(v["rho"]), _ptr(v["Yt"]), _ptr(v["Ft"]),
                                     _ptr(v["Q"]), *[C.cast(C.byref(x), _dp) for x in s])
        )  # fmt: skip
        out = dict(v, G=s[0].value, beta=s[1].value, sigma2=s[2].value, noise_var=s[3].value)
        if with_C:
            out["C"] = Cm
        if getattr(self, "trend", TREND_CONSTANT) != TREND_CONSTANT:  # p > 1: Ft, Q (N, p), G (p, p), beta (p,)
            p = self.trend_size()
            est = getattr(self, "estimate_trend", False)
            Ft, Q, G, beta = np.zeros((N, p)), np.zeros((N, p)), np.zeros((p, p)), np.zeros(p)
            self._check(self._lib.bogp_get_trend_state(self._h, _ptr(Ft) if est else None, _ptr(Q) if est else None,
                                                       _ptr(G) if est else None, _ptr(beta)))  # fmt: skip
            out.update(Ft=Ft, Q=Q, G=G, beta=beta)
        return out

    # -- candidates ---------------------------------------------------------------------------------
    def upload_candidates(self, Xs, lazy=False):
        """Host candidates (M, d).  `lazy=True`: only the head is copied now, the rest travels chunk by chunk beside the kernels of the
        NEXT predict / sweep / sweep_topk call (bogp_candidates_upload_lazy) -- the array is kept alive by the engine until then, and
        the caller must not modify it before that call returns."""
        Xs = _f64(Xs)
        if Xs.ndim != 2 or Xs.shape[1] != self.d:
            raise ValueError("candidates must have shape (M, %d)" % self.d)
        fn = self._lib.bogp_candidates_upload_lazy if lazy else self._lib.bogp_candidates_upload
        self._check(fn(self._h, _ptr(Xs), Xs.shape[0]))
        self.M = Xs.shape[0]
        self._last_q, self._last_topk = -1, (-1, -1)  # winners of an earlier sweep refer to other rows
        self._keep = Xs if lazy else None

    def bind_candidates(self, device_ptr: int, M: int, owner=None):
        """Adopt caller-owned device memory (M x d float64 row-major), e.g. a torch tensor's data_ptr()."""
        self._check(self._lib.bogp_candidates_bind(self._h, C.c_void_p(int(device_ptr)), int(M)))
        self.M = int(M)
        self._last_q, self._last_topk = -1, (-1, -1)
        self._keep = owner

    def generate_candidates(self, lo, hi, M: int, seed: int = 0, first_row: int = 0, method: str = "uniform",
                            n_total: Optional[int] = None, sobol_sv: Optional[np.ndarray] = None, maximin: int = 5):
        """M points in the box [lo, hi] drawn ON the device -- no host sampling, no H2D copy.  `method` follows
        RealSpace._sample (search_space.py:742-754): "uniform" (Philox4x32-10 stream `seed`, rows [first_row,
        first_row + M)); "LHS" (rows [first_row, first_row + M) of an `n_total`-point Latin hypercube, default M);
        "sobol" (points first_row + 1 ... of the unscrambled sequence -- the reference skips point 0 -- for the
        direction numbers `sobol_sv` (d x bits), default scipy's); "LHS-maximin" (the best of `maximin` hypercubes by
        minimum pairwise distance, pyDOE's criterion; (distance, trial) left in `self.last_maximin`)."""
        lo, hi = _f64(lo).ravel(), _f64(hi).ravel()
        if len(lo) != self.d or len(hi) != self.d:
            raise ValueError("bounds must have %d entries" % self.d)
        useed = C.c_uint64(int(seed) & (2**64 - 1))
        if method == "uniform":
            rc = self._lib.bogp_candidates_generate(self._h, _ptr(lo), _ptr(hi), int(M), useed, int(first_row))
        elif method == "LHS":
            n_total = int(M) + int(first_row) if n_total is None else int(n_total)
            rc = self._lib.bogp_candidates_generate_lhs(self._h, _ptr(lo), _ptr(hi), int(M), useed, int(first_row), n_total)
        elif method == "LHS-maximin":  # pyDOE's criterion="maximin": the reference's own "LHS" (search_space.py:751)
            if int(first_row) != 0 or (n_total is not None and int(n_total) != int(M)):
                raise NotImplementedError("the maximin criterion needs the whole design on one device (no row shards)")
            dist, it = C.c_double(), C.c_int()
            rc = self._lib.bogp_candidates_generate_lhs_maximin(self._h, _ptr(lo), _ptr(hi), int(M), useed, int(maximin),
                                                                 C.cast(C.byref(dist), _dp), C.cast(C.byref(it), _ip))  # fmt: skip
            self.last_maximin = (dist.value, it.value)
        elif method == "sobol":
            sv = sobol_direction_numbers(self.d) if sobol_sv is None else sobol_sv
            sv = np.ascontiguousarray(sv, dtype=np.uint64)
            if sv.ndim != 2 or sv.shape[0] != self.d:
                raise ValueError("sobol_sv must be (d, bits)")
            rc = self._lib.bogp_candidates_generate_sobol(self._h, _ptr(lo), _ptr(hi), int(M), int(first_row) + 1,
                                                          sv.ctypes.data_as(C.POINTER(C.c_uint64)), int(sv.shape[1]))  # fmt: skip
        else:
            raise ValueError("method must be 'uniform', 'LHS' or 'sobol'")
        self._check(rc)
        self.M = int(M)
        self._last_q, self._last_topk = -1, (-1, -1)
        self._keep = None

    def set_candidate_transform(self, scales=None, precisions=None, lo=None, hi=None):
        """Post-processing of device-generated candidates as RealSpace._sample does it (search_space.py:754): `scales` per
        dimension ("linear" | "log" | "log10" | "logit" | "bilog": the design is drawn in the transformed box and mapped
        back), `precisions` (decimals or None) with the variables' own bounds `lo`, `hi` for the clip.  No arguments:
        plain designs again."""
        if scales is None and precisions is None:
            self._check(self._lib.bogp_candidates_set_transform(self._h, None, None, None, None))
            return
        d = self.d
        sc = np.ascontiguousarray([SCALES[s] for s in (scales if scales is not None else [None] * d)], dtype=np.int32)
        pr = np.ascontiguousarray([-1 if p is None else int(p) for p in (precisions if precisions is not None else [None] * d)], dtype=np.int32)
        if len(sc) != d or len(pr) != d:
            raise ValueError("scales / precisions must have %d entries" % d)
        lo_ = _f64(lo).ravel() if lo is not None else None
        hi_ = _f64(hi).ravel() if hi is not None else None
        self._check(self._lib.bogp_candidates_set_transform(self._h, sc.ctypes.data_as(_ip), pr.ctypes.data_as(_ip), _ptr(lo_), _ptr(hi_)))

    def min_pairwise_distance(self) -> float:
        """min over pairs of the Euclidean distance between the current candidates (scipy's pdist(...).min())."""
        out = C.c_double()
        self._check(self._lib.bogp_candidates_min_pdist2(self._h, C.cast(C.byref(out), _dp)))
        return float(np.sqrt(out.value))

    def read_candidates(self, rows) -> np.ndarray:
        rows = np.ascontiguousarray(rows, dtype=np.int64).ravel()
        out = np.empty((len(rows), self.d))
        self._check(self._lib.bogp_candidates_read(self._h, rows.ctypes.data_as(_lp), len(rows), _ptr(out)))
        return out

    # -- posterior / sweep ----------------------------------------------------------------------------
    def predict(self, eval_MSE=True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        mu = np.empty(self.M)
        mse = np.empty(self.M) if eval_MSE else None
        self._check(self._lib.bogp_predict(self._h, _ptr(mu), _ptr(mse)))
        return mu, mse

    def sweep(self, acq: Sequence[Tuple[int, float]], plugin: float, minimize=True, return_values=False, local_result=True):
        """Posterior + q criteria + argmax over the current candidates: (best_val (q,), best_idx (q,)[, values (q, M)]).
        local_result=False only queues the sweep (no host wait, returns None): its winners stay on the device for the
        exchange_argmax() call that follows."""
        q = len(acq)
        ids = np.ascontiguousarray([a for a, _ in acq], dtype=np.int32)
        pars = _f64([float(p) if p is not None else 0.0 for _, p in acq])
        self._last_q, self._last_topk = -1, (-1, -1)
        if not local_result:
            self._check(self._lib.bogp_sweep(self._h, q, ids.ctypes.data_as(_ip), _ptr(pars), float(plugin), int(bool(minimize)),
                                             None, None, None))  # fmt: skip
            self._last_q = q
            return None
        best = np.empty(q)
        idx = np.empty(q, dtype=np.int64)
        vals = np.empty((q, self.M)) if return_values else None
        self._check(
            self._lib.bogp_sweep(self._h, q, ids.ctypes.data_as(_ip), _ptr(pars), float(plugin), int(bool(minimize)),
                                 _ptr(best), idx.ctypes.data_as(_lp), _ptr(vals))
        )  # fmt: skip
        self._last_q = q
        return (best, idx, vals) if return_values else (best, idx)

    def sweep_topk(self, acq: Sequence[Tuple[int, float]], plugin: float, minimize=True, k: int = 1):
        """k best candidates per criterion: (values (q, k), indices (q, k)); rank 0 is the argmax."""
        q = len(acq)
        ids = np.ascontiguousarray([a for a, _ in acq], dtype=np.int32)
        pars = _f64([float(p) if p is not None else 0.0 for _, p in acq])
        best = np.empty((q, k))
        idx = np.empty((q, k), dtype=np.int64)
        self._last_q, self._last_topk = -1, (-1, -1)
        self._check(
            self._lib.bogp_sweep_topk(self._h, q, ids.ctypes.data_as(_ip), _ptr(pars), float(plugin), int(bool(minimize)),
                                      int(k), _ptr(best), idx.ctypes.data_as(_lp))
        )  # fmt: skip
        self._last_topk = (q, int(k))
        return best, idx

    def gradient(self, x):
        x = _f64(x).ravel()
        if len(x) != self.d:
            raise Exception("x does not have the right size!")
        dmu, dmse = np.empty(self.d), np.empty(self.d)
        self._check(self._lib.bogp_gradient(self._h, _ptr(x), _ptr(dmu), _ptr(dmse)))
        return dmu, dmse

    def point_eval(self, x, acq: Sequence[Tuple[int, float]] = (), plugin: float = 0.0, minimize: bool = True):
        """mu, mse, dmu (d,), dmse (d,), criterion values (q,) at ONE point with one device round trip
        (bogp_point_eval: what `criterion(x, return_dx=True)` needs).  Constant trend basis.
        This is the call the reference's L-BFGS-B loop makes thousands of times per ask(): the ctypes argument objects are
        built once per (d, criteria) and re-used, so that the host side of a call is a few microseconds."""
        key = (self.d, tuple(acq))
        c = self.__dict__.get("_pe_cache")
        if c is None or c[0] != key:
            q = len(acq)
            ids = np.ascontiguousarray([a for a, _ in acq], dtype=np.int32)
            pars = np.ascontiguousarray([p for _, p in acq], dtype=np.float64)
            xb, out = np.empty(self.d), np.empty(2 + 2 * self.d + max(q, 1))
            po = out.ctypes.data
            c = (key, q, xb, out, ids, pars, _ptr(xb), ids.ctypes.data_as(_ip) if q else None, _ptr(pars) if q else None,
                 C.cast(po, _dp), C.cast(po + 8, _dp), C.cast(po + 16, _dp), C.cast(po + 16 + 8 * self.d, _dp),
                 C.cast(po + 16 + 16 * self.d, _dp) if q else None)  # fmt: skip
            self._pe_cache = c
        _, q, xb, out, _, _, px, pids, ppars, pmu, pmse, pdmu, pdmse, pvals = c
        x = np.asarray(x, dtype=np.float64)
        if x.size != self.d:
            raise Exception("x does not have the right size!")
        xb[:] = x.ravel()
        rc = self._lib.bogp_point_eval(self._h, px, q, pids, ppars, float(plugin), 1 if minimize else 0, pmu, pmse, pdmu, pdmse, pvals)
        if rc:
            self._check(rc)
        d = self.d
        return float(out[0]), float(out[1]), out[2 : 2 + d].copy(), out[2 + d : 2 + 2 * d].copy(), out[2 + 2 * d : 2 + 2 * d + q].copy()

    def point_eval_batch(self, Xb, acq: Sequence[Tuple[int, float]] = (), plugin: float = 0.0, minimize: bool = True,
                         return_dx: bool = True):
        """(mu (B,), mse (B,), dmu (B, d), dmse (B, d), values (B, q), dvalues (B, q, d) or None) at the B rows of `Xb` in ONE
        device round trip (bogp_point_eval_batch): the criteria's input-gradients are the reference's `return_dx` chain rule
        evaluated on the device.  Constant trend basis."""
        Xb = _f64(Xb)
        if Xb.ndim != 2 or Xb.shape[1] != self.d:
            raise Exception("x does not have the right size!")
        B, q = Xb.shape[0], len(acq)
        ids = np.ascontiguousarray([a for a, _ in acq], dtype=np.int32)
        pars = np.ascontiguousarray([p for _, p in acq], dtype=np.float64)
        mu, mse = np.empty(B), np.empty(B)
        dmu, dmse = np.empty((B, self.d)), np.empty((B, self.d))
        vals = np.empty((B, max(q, 1)))
        dvals = np.empty((B, max(q, 1), self.d)) if (return_dx and q) else None
        self._check(
            self._lib.bogp_point_eval_batch(self._h, _ptr(Xb), B, q, ids.ctypes.data_as(_ip) if q else None, _ptr(pars) if q else None,
                                            float(plugin), int(bool(minimize)), _ptr(mu), _ptr(mse), _ptr(dmu), _ptr(dmse),
                                            _ptr(vals) if q else None, _ptr(dvals))
        )  # fmt: skip
        return mu, mse, dmu, dmse, vals[:, :q], (dvals[:, :q] if dvals is not None else None)

    def polish(self, X0, lo, hi, acq: Tuple[int, float], plugin: float = 0.0, minimize: bool = True, max_evals: int = 50,
               pgtol: float = 1e-8, factr: float = 1e6):
        """Lock-step multi-start local maximisation of one criterion inside the box (bogp_polish): (points (B, d), values (B,),
        evaluations used (B,)); values[i] >= the criterion at X0[i]."""
        X0 = _f64(X0)
        if X0.ndim != 2 or X0.shape[1] != self.d:
            raise Exception("x does not have the right size!")
        lo, hi = _f64(lo).ravel(), _f64(hi).ravel()
        if len(lo) != self.d or len(hi) != self.d:
            raise ValueError("lo / hi must have d entries")
        B = X0.shape[0]
        Xo, fo, ne = np.empty((B, self.d)), np.empty(B), np.zeros(B, dtype=np.int32)
        self._check(
            self._lib.bogp_polish(self._h, _ptr(X0), B, _ptr(lo), _ptr(hi), int(acq[0]), float(acq[1]), float(plugin),
                                  int(bool(minimize)), int(max_evals), float(pgtol), float(factr), _ptr(Xo), _ptr(fo),
                                  ne.ctypes.data_as(_ip))
        )  # fmt: skip
        return Xo, fo, ne

    def hessian(self, x) -> np.ndarray:
        """(d, d) Hessian of the posterior mean at x (squared exponential; constant / linear trend)."""
        x = _f64(x).ravel()
        if len(x) != self.d:
            raise Exception("x does not have the right size!")
        H = np.empty((self.d, self.d))
        self._check(self._lib.bogp_hessian(self._h, _ptr(x), _ptr(H)))
        return H

    def prior_corr(self, X1) -> np.ndarray:
        """(n1, n1) correlation matrix of the rows of X1 at the committed theta."""
        X1 = _f64(X1)
        if X1.ndim != 2 or X1.shape[1] != self.d:
            raise ValueError("X1 must have shape (n1, %d)" % self.d)
        R = np.empty((X1.shape[0], X1.shape[0]))
        self._check(self._lib.bogp_prior_corr(self._h, _ptr(X1), X1.shape[0], _ptr(R)))
        return R

    def gradient_batch(self, Xb):
        """(d mu / dx, d MSE / dx) at B points: two (B, d) arrays."""
        Xb = _f64(Xb)
        if Xb.ndim != 2 or Xb.shape[1] != self.d:
            raise Exception("x does not have the right size!")
        B = Xb.shape[0]
        dmu, dmse = np.empty((B, self.d)), np.empty((B, self.d))
        self._check(self._lib.bogp_gradient_batch(self._h, _ptr(Xb), B, _ptr(dmu), _ptr(dmse)))
        return dmu, dmse

    # -- multi-GPU exchange (bogp_comm.hip; one process per GPU) ---------------------------------------------
    def comm_init(self, comm_id: bytes, rank: int, world: int):
        """RCCL communicator over `world` handles (one per process / GPU); `comm_id` = comm_unique_id() of ONE rank."""
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("comm_id must be %d bytes" % COMM_ID_BYTES)
        self._check(self._lib.bogp_comm_init(self._h, comm_id, int(rank), int(world)))
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_attach(self, nccl_comm_ptr: int):
        self._check(self._lib.bogp_comm_attach(self._h, C.c_void_p(int(nccl_comm_ptr))))
        self.comm_rank, self.comm_world = self.comm_info()

    def comm_info(self) -> Tuple[int, int]:
        """(rank, world); world == 0 when the handle has no communicator."""
        r, w = C.c_int(), C.c_int()
        self._check(self._lib.bogp_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_destroy(self):
        self._check(self._lib.bogp_comm_destroy(self._h))
        self.comm_rank, self.comm_world = 0, 0

    def exchange_argmax(self, q: int, index_offset: int = 0, with_points: bool = True):
        """After sweep(): the global winners over all ranks' shards, identical on every rank:
        (best_val (q,), best_global_idx (q,), best_x (q, d) or None).  ONE ncclAllGather on device records."""
        if int(q) != self.__dict__.get("_last_q", -1):  # the C side packs ITS q records: a mismatch would under- or overrun these buffers
            raise ValueError("exchange_argmax(q=%d) does not follow a sweep of q = %d criteria on the current candidates" % (q, self.__dict__.get("_last_q", 0)))
        best, gidx = np.empty(q), np.empty(q, dtype=np.int64)
        x = np.empty((q, self.d)) if with_points else None
        self._check(self._lib.bogp_exchange_argmax(self._h, int(index_offset), _ptr(best), gidx.ctypes.data_as(_lp), _ptr(x)))
        return best, gidx, x

    def exchange_topk(self, q: int, k: int, index_offset: int = 0, with_points: bool = True):
        """After sweep_topk(): (values (q, k), global indices (q, k), points (q, k, d) or None), identical on every rank."""
        if (int(q), int(k)) != self.__dict__.get("_last_topk", (-1, -1)):
            raise ValueError("exchange_topk(q=%d, k=%d) does not follow a sweep_topk of that shape on the current candidates" % (q, k))
        best, gidx = np.empty((q, k)), np.empty((q, k), dtype=np.int64)
        x = np.empty((q, k, self.d)) if with_points else None
        self._check(self._lib.bogp_exchange_topk(self._h, int(index_offset), _ptr(best), gidx.ctypes.data_as(_lp), _ptr(x)))
        return best, gidx, x

    def last_timing(self) -> dict:
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int()
        cast = lambda x: C.cast(C.byref(x), _dp)  # noqa: E731
        self._check(self._lib.bogp_last_timing(self._h, cast(a), cast(b), cast(c), C.cast(C.byref(n), _ip)))
        return dict(corr_ms=a.value, contract_ms=b.value, acquisition_ms=c.value, n_chunks=n.value)

    def selftest_gemm(self, A, B, C_in=None, ta=False, tb=False, alpha=1.0, beta=0.0, tri=0, split=True):
        """alpha op(A) op(B) + beta C through k_gemm64 (kernels_gemm.hip); A, B, C_in Fortran-ordered 2-D float64 arrays."""
        A = np.asfortranarray(A, dtype=np.float64)
        B = np.asfortranarray(B, dtype=np.float64)
        m, k = (A.shape[1], A.shape[0]) if ta else A.shape
        k2, n = (B.shape[1], B.shape[0]) if tb else B.shape
        if k != k2:
            raise ValueError("inner dimensions differ: %d vs %d" % (k, k2))
        out = np.zeros((m, n), order="F") if C_in is None else np.asfortranarray(C_in, dtype=np.float64).copy(order="F")
        self._check(self._lib.bogp_selftest_gemm(self._h, int(ta), int(tb), m, n, k, float(alpha), _ptr(A), A.shape[0], _ptr(B), B.shape[0],
                                                 float(beta), _ptr(out), out.shape[0], int(tri), int(split)))
        return out

    def flops_per_candidate(self) -> float:
        return float(self._lib.bogp_flops_per_candidate(self._h))

    def selftest_profile(self, what, arg, kernel=0, pexp=0.0):
        """`what` (SELFTEST_*) of every entry of `arg` on the device: the radial profile of `kernel`, K_pexp or 1 / Gamma exactly as the
        producers evaluate them per pair (csrc/bogp_device.h).  The *_PAIRS selectors take an (n, 2) array of (order, argument) rows."""
        arg = np.ascontiguousarray(arg, dtype=np.float64)
        n = arg.shape[0] if what >= SELFTEST_BESSEL_K_PAIRS else arg.size
        if what >= SELFTEST_BESSEL_K_PAIRS and (arg.ndim != 2 or arg.shape[1] != 2):
            raise ValueError("the *_PAIRS selectors take an (n, 2) array")
        out = np.empty(n)
        self._check(self._lib.bogp_selftest_profile(self._h, int(what), int(kernel), float(pexp), _ptr(arg), n, _ptr(out)))
        return out
